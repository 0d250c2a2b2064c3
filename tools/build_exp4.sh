# experimental variant of the product library (round 4): tools/build_exp4.sh name "<flags for every v3 kernel file and the host side>"
# rebuilds demod_kernel_rot.hip, demod_kernel_rotp.hip, demod_api.cpp and demod_host.cpp with the extra flags and links them with the
# product's other objects into gpurun_exp/<name>.so (A/B on one box: MDEMOD_LIB_PATH, tools/ab_bench.sh)
set -e
name=$1; flags=$2
HC=/opt/rocm/bin/hipcc
COMMON="-O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function --offload-arch=gfx950 -Iinclude -x hip -c"
D=/tmp/exp4_$name
mkdir -p gpurun_exp $D
( $HC $COMMON -fno-slp-vectorize -Wno-inline-asm -mllvm -amdgpu-sched-strategy=max-ilp $flags meteor_demod_amd/csrc/demod_kernel_rot.hip -o $D/rot.o --save-temps=obj 2>&1 | grep -v "hip-link\|warning\|note:\|^$" || true ) &
( $HC $COMMON -fno-slp-vectorize -Wno-inline-asm $flags meteor_demod_amd/csrc/demod_kernel_rotp.hip -o $D/rotp.o --save-temps=obj 2>&1 | grep -v "hip-link\|warning\|note:\|^$" || true ) &
( $HC $COMMON $flags meteor_demod_amd/csrc/demod_api.cpp -o $D/api.o 2>&1 | grep -v "hip-link\|warning\|note:\|^$" || true ) &
( $HC $COMMON $flags meteor_demod_amd/csrc/demod_host.cpp -o $D/host.o 2>&1 | grep -v "hip-link\|warning\|note:\|^$" || true ) &
wait
objs=$(ls meteor_demod_amd/lib/*.o | grep -v "demod_kernel_rot.o\|demod_kernel_rotp.o\|demod_api.o\|demod_host.o" | tr '\n' ' ')
$HC -shared -fPIC -pthread --offload-arch=gfx950 -o gpurun_exp/$name.so $objs $D/rot.o $D/rotp.o $D/api.o $D/host.o
for k in demod_kernel_rotILi16ELi0ELi14 demod_kernel_rotILi16ELi1ELi6; do python3 tools/isa_loop_stats.py $D/demod_kernel_rot-hip-amdgcn-amd-amdhsa-gfx950.s $k | grep -v whole | tr '\n' ' '; echo; done
python3 tools/isa_loop_stats.py $D/demod_kernel_rotp-hip-amdgcn-amd-amdhsa-gfx950.s WIDE_16_0_ks109 | grep -v whole | tr '\n' ' '; echo
ls -la gpurun_exp/$name.so
