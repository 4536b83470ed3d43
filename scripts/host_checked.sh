#!/bin/bash
# The HOST side of libssd_hip.so with run-time checks: abi / weights / plan / stages -- weight packing, source tables, arena
# layout, plan construction, the ABI's argument checks -- compiled with UBSan (-fsanitize=undefined, no recovery: bounds of
# fixed arrays, signed overflow, misaligned / null accesses, bad shifts, float -> int overflow) and -D_GLIBCXX_ASSERTIONS
# (every std::vector / std::string / std::array index checked); the kernel files are the shipped objects, the device code is
# not instrumented.  (AddressSanitizer is not possible beside the HIP runtime on this pool: with ASan's shadow mapping in
# place hipInit aborts, and there is no ASan build of the ROCm runtime in the image.)
#   build (here, needs hipcc):   scripts/host_checked.sh build
#   run (GPU box):               scripts/host_checked.sh run [pytest args]
set -e
cd "$(dirname "$0")/.."
C=single-shot-detector_amd/csrc
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)")
if [ "$1" = build ]; then
    python -c "import ssd_amd; ssd_amd.build()"            # the shipped kernel objects, fresh
    mkdir -p $C/build/chk
    for f in abi weights plan stages; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -ffp-contract=off -fPIC -std=c++17 -fsanitize=undefined \
            -fno-sanitize-recover=undefined -fno-gpu-sanitize -D_GLIBCXX_ASSERTIONS -fno-omit-frame-pointer \
            -c $C/$f.hip -o $C/build/chk/$f.hip.o &
    done
    wait
    objs=""
    for f in abi weights plan stages; do objs="$objs $C/build/chk/$f.hip.o"; done
    for f in igemm igemm_lat igemm16 dwpw_stream sn_pw front elementwise postprocess; do objs="$objs $C/build/ship/$f.hip.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=undefined -fno-gpu-sanitize -shared-libsan \
        -Wl,--version-script=$C/exports.map -o $C/libssd_hip_chk.so $objs
    echo built $C/libssd_hip_chk.so
    exit 0
fi
shift || true
export LD_LIBRARY_PATH=$RT${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
exec python scripts/host_checked.py "$@"
