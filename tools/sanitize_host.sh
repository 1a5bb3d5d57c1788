#!/bin/bash
# The HOST side of the engine's translation unit (track and direct planning, frame facts, lane packing helpers -- everything the host-only
# C-ABI views reach: speechPlayer_planTracks / _planTracksFacts, speechPlayer_planDirect, speechPlayer_frameFacts, and the frame producer's
# compact form on the engine's worker pool) under AddressSanitizer + UBSan, no GPU needed:
# the device code is compiled as usual, the host code with -Xarch_host -fsanitize=address,undefined, and the planning tests run against
# that library (GPU AddressSanitizer is not available on this pool; the frame producer has its own sanitizer test in the CPU suite).
#   bash tools/sanitize_host.sh            (about three minutes, most of it the compile)
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/speechplayer_asan
mkdir -p "$OUT"
/opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Xarch_host -fsanitize=address,undefined \
    -Xarch_host -fno-omit-frame-pointer -c nvspeechplayer_amd/csrc/klatt_engine.hip -o "$OUT/klatt_engine.o"
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -fPIC -std=c++17 -fsanitize=address,undefined -c nvspeechplayer_amd/csrc/frame_producer.cpp -o "$OUT/frame_producer.o"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -shared-libsan -Wl,-rpath,/opt/rocm/lib \
    -o "$OUT/libspeechPlayer_asan.so" "$OUT/klatt_engine.o" "$OUT/frame_producer.o"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
for threads in 8 2; do
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 SPEECHPLAYER_PLAN_THREADS=$threads SPEECHPLAYER_LIB="$OUT/libspeechPlayer_asan.so" \
      python -m pytest tests/test_track_planning.py tests/test_direct_planning.py tests/test_host_logic.py tests/test_ipa_producer.py -x -q -k "not sanitizers and not product_does_not"
done
