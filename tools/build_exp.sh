# build an experimental variant of the product library into gpurun_exp/<name>.so: tools/build_exp.sh name "<extra flags for demod_kernel_rw std part>" ["<extra flags for lat>"]
set -e
name=$1; rwflags=$2
HC=/opt/rocm/bin/hipcc
COMMON="-O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function --offload-arch=gfx950 -Iinclude -x hip -c"
mkdir -p gpurun_exp /tmp/exp_$name
$HC $COMMON -fno-slp-vectorize -DMDEMOD_RW_PART=1 -mllvm -amdgpu-sched-strategy=max-ilp $rwflags meteor_demod_amd/csrc/demod_kernel_rw.hip -o /tmp/exp_$name/rw_std.o 2>&1 | grep -v hip-link || true
objs=$(ls meteor_demod_amd/lib/*.o | grep -v demod_kernel_rw_std.o | tr '\n' ' ')
$HC -shared -fPIC -pthread --offload-arch=gfx950 -o gpurun_exp/$name.so $objs /tmp/exp_$name/rw_std.o
ls -la gpurun_exp/$name.so
