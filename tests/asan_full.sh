# AddressSanitizer pass over ALL host code of the library (also the setup code inside the .hip files: row blocks, SpGEMM symbolic phase,
# multigrid and halo setup) -- device code is compiled without instrumentation (-fno-gpu-sanitize).
#   bash tests/asan_full.sh build          (here: cross-compiles femus_amd/lib/asan/libfemus_hip.so, which travels to the GPU box)
#   bash tests/asan_full.sh run [pytest args]     (on the GPU box: the -m gpu suite with that library preloaded)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/femus_amd/lib/asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
if [ "$1" = "build" ]; then
  mkdir -p $OUT/obj
  for f in $ROOT/femus_amd/csrc/*.hip $ROOT/femus_amd/csrc/*.cpp; do
    o=$OUT/obj/$(basename $f).o
    if [ ! -f $o ] || [ $f -nt $o ] || [ $ROOT/femus_amd/csrc/fh_internal.h -nt $o ]; then
      echo "asan: $(basename $f)"
      /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -x hip -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer \
          -c $f -o $o &
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -o $OUT/libfemus_hip.so $OUT/obj/*.o \
      -lpthread -L/opt/rocm/lib -lrccl
  ls -la $OUT/libfemus_hip.so
else
  shift
  cd $ROOT
  FEMUS_HIP_LIBRARY=$OUT/libfemus_hip.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0:detect_odr_violation=0 \
      python -m pytest tests -q -m gpu -s "$@"
fi
