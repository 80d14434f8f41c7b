#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of liblsrecon -- the host twins, the estimator twins,
# the blosc frame walker, the device codecs' twins (the kernels' own zstd source compiled for the host), and the argument checks / launch planning in front of every kernel -- on the CPU, where
# sanitizers run (GPU ASan is not available on the pool):
#   bash tools/host_sanitize.sh [scratch-dir] [fuzz-seconds]        # from the repo root; no GPU needed (or wanted)
# The sources are copied to a scratch directory and every translation unit's host side is built there with
# -fsanitize=address,undefined (the device code is untouched).  Then, against that library (LSR_LIBRARY, the ASan runtime
# preloaded into python): the CPU tests that drive the host code, damaged blosc frames (tools/fuzz_blosc.py), random
# arguments through the host twins (tools/fuzz_host_args.py) and -- on a machine without a GPU, where nothing can
# launch -- random arguments through every device entry point (tools/fuzz_device_args.py).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/lsr_asan_all}
SECS=${2:-60}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p "$OUT/shrimpy_amd/csrc" "$OUT/include"
cp "$R"/shrimpy_amd/csrc/*.hip "$R"/shrimpy_amd/csrc/*.hpp "$R"/shrimpy_amd/csrc/Makefile "$OUT/shrimpy_amd/csrc/"
cp "$R/include/lsrecon.h" "$OUT/include/"
cd "$OUT/shrimpy_amd/csrc"
make -j8 EXTRA="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -O1" liblsrecon.so > build.log 2>&1 \
  || { tail -20 build.log; exit 1; }
$HIPCC -shared -fPIC --offload-arch=gfx950 *.o -fsanitize=address,undefined -shared-libsan -o liblsrecon.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$R"
export LSR_LIBRARY="$OUT/shrimpy_amd/csrc/liblsrecon.so" LD_PRELOAD="$RT"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_odr_violation=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_host_twins.py tests/test_dynatrack_host.py tests/test_zarr_codecs.py tests/test_host.py \
       tests/test_preprocessing_mirror.py tests/test_io_cli.py tests/test_device_codec.py -q -m "not gpu" -p no:cacheprovider | tail -2
python tools/fuzz_blosc.py --seconds "$SECS"
python tools/fuzz_device_codec.py --seconds "$SECS"
python tools/fuzz_host_args.py --seconds "$SECS"
python tools/fuzz_device_args.py --seconds "$SECS"
