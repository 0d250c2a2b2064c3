# experimental variant of the product library with another build of the v3 packed-window kernel: tools/build_exp_rotp.sh name "<extra hipcc flags>" ["<env for gen_rotpk_asm.py>"]
set -e
name=$1; flags=$2; genenv=$3
HC=/opt/rocm/bin/hipcc
COMMON="-O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function --offload-arch=gfx950 -Iinclude -x hip -c"
mkdir -p gpurun_exp /tmp/exp_$name
env $genenv python3 meteor_demod_amd/csrc/gen_rotpk_asm.py > /tmp/exp_$name/rotpk_asm.h
$HC $COMMON -fno-slp-vectorize -Wno-inline-asm -DROTPK_ASM_HEADER='"/tmp/exp_'$name'/rotpk_asm.h"' $flags meteor_demod_amd/csrc/demod_kernel_rotp.hip -o /tmp/exp_$name/rotp.o --save-temps=obj 2>&1 | grep -v "hip-link\|warning\|note:\|^$" || true
cp /tmp/exp_$name/demod_kernel_rotp-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/exp_$name/rotp.s
objs=$(ls meteor_demod_amd/lib/*.o | grep -v demod_kernel_rotp.o | tr '\n' ' ')
$HC -shared -fPIC -pthread --offload-arch=gfx950 -o gpurun_exp/$name.so $objs /tmp/exp_$name/rotp.o
python3 tools/isa_loop_stats.py /tmp/exp_$name/rotp.s demod_kernel_rotp_WIDE_16_0 | grep -v whole | tr '\n' ' '; echo
ls -la gpurun_exp/$name.so
