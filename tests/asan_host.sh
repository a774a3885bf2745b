# AddressSanitizer + UBSan pass over the HOST-side sources of the library (mesh, planner, expression parser, FE tables, writers) through
# the CPU test suite: the .cpp files are rebuilt with -fsanitize=address,undefined into /tmp/femus_asan/libfemus_hip.so (the .hip objects
# are taken from the normal build), and pytest runs with that library preloaded.   bash tests/asan_host.sh [pytest args]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/femus_asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p $OUT
make -s -C $ROOT/femus_amd/csrc -j8
objs=""
for f in $ROOT/femus_amd/csrc/*.cpp; do
  o=$OUT/$(basename $f).o
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/include -fsanitize=address,undefined \
      -fno-sanitize=vptr -fno-omit-frame-pointer -c $f -o $o
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $OUT/libfemus_hip.so \
    $(ls $ROOT/femus_amd/lib/obj/*.o | grep -v "\.cpp\.o$") $objs -lpthread -L/opt/rocm/lib -lrccl
shift 0
cd $ROOT
FEMUS_HIP_LIBRARY=$OUT/libfemus_hip.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=0:halt_on_error=0 \
    python -m pytest tests -q -m "not gpu" -x "$@"
