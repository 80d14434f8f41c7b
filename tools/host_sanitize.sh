#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST code of liblsrecon (the host twins, the estimator twins,
# the blosc frame walker, the C-ABI glue) -- on the CPU, where sanitizers run (GPU ASan is not available on the pool):
#   bash tools/host_sanitize.sh [outdir]        # from the repo root, after `make -C shrimpy_amd/csrc`
# The four host-only translation units are rebuilt with -fsanitize=address,undefined and linked with the kernels'
# objects of the normal build into <outdir>/liblsrecon.so; the CPU tests that drive that code then run against it
# (LSR_LIBRARY picks the library, the ASan runtime is preloaded into python).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/lsr_asan}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p "$OUT"
cd "$R/shrimpy_amd/csrc"
HOST_TUS="api host_twins estimators_host blosc_frame"
for f in $HOST_TUS; do
  $HIPCC -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xarch_host -mfma \
         -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -c $f.hip -o "$OUT/$f.o"
done
OBJS=$(ls *.o | grep -v -E "^(api|host_twins|estimators_host|blosc_frame)\.o$")
$HIPCC -shared -fPIC --offload-arch=gfx950 $OBJS "$OUT"/*.o -fsanitize=address,undefined -shared-libsan -o "$OUT/liblsrecon.so"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$R"
LSR_LIBRARY="$OUT/liblsrecon.so" LD_PRELOAD="$RT" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_odr_violation=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest tests/test_host_twins.py tests/test_dynatrack_host.py tests/test_zarr_codecs.py tests/test_host.py \
       tests/test_preprocessing_mirror.py tests/test_io_cli.py -q -m "not gpu" -p no:cacheprovider
