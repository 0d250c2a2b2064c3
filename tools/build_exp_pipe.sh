# experimental variant of the host-buffer pipeline (round 5): tools/build_exp_pipe.sh name "<flags>"
# rebuilds csrc/host_pipe.cpp with the extra flags (e.g. -DMDEMOD_PIPE_TRACE -DMDEMOD_PIPE_KMAX=16 -DMDEMOD_PIPE_MINSAMP=4096
# -DMDEMOD_PIPE_SUBBYTES=33554432) and links it with the product's other objects into gpurun_exp/<name>.so (MDEMOD_LIB_PATH)
set -e
name=$1; flags=$2
HC=/opt/rocm/bin/hipcc
mkdir -p gpurun_exp /tmp/exp_pipe_$name
$HC -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function --offload-arch=gfx950 -Iinclude -x hip -c $flags meteor_demod_amd/csrc/host_pipe.cpp -o /tmp/exp_pipe_$name/host_pipe.o
objs=$(ls meteor_demod_amd/lib/*.o | grep -v "host_pipe.o" | tr '\n' ' ')
$HC -shared -fPIC -pthread --offload-arch=gfx950 -o gpurun_exp/$name.so $objs /tmp/exp_pipe_$name/host_pipe.o
ls -la gpurun_exp/$name.so
