# experimental variant of the product library with another build of the latency kernel: tools/build_exp_lat.sh name "<extra hipcc flags>"
set -e
name=$1; flags=$2
HC=/opt/rocm/bin/hipcc
COMMON="-O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function --offload-arch=gfx950 -Iinclude -x hip -c"
mkdir -p gpurun_exp /tmp/exp_$name
$HC $COMMON -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp $flags meteor_demod_amd/csrc/demod_kernel_lat.hip -o /tmp/exp_$name/lat.o 2>&1 | grep -v "hip-link\|warning\|note:\|^$" || true
objs=$(ls meteor_demod_amd/lib/*.o | grep -v demod_kernel_lat.o | tr '\n' ' ')
$HC -shared -fPIC -pthread --offload-arch=gfx950 -o gpurun_exp/$name.so $objs /tmp/exp_$name/lat.o
ls -la gpurun_exp/$name.so
