#!/bin/bash
# Round 6: the product sources (kernels' C++ + the whole host runtime) under AddressSanitizer + UndefinedBehaviorSanitizer — the
# one-lane emulation build with both, the 64-lane build (fibers: ASan would need fiber annotations) with UBSan.  Any report aborts.
#   bash profiles/tools/r10_sanitizers.sh > profiles/r10_sanitizers.txt 2>&1     (run from the repo root)
SRC="-x c++ era-zk_evm_amd/csrc/zkw_kernels.hip -x c++ era-zk_evm_amd/csrc/zkw_commit.hip -x c++ era-zk_evm_amd/csrc/zkw_blake2s.hip -x c++ era-zk_evm_amd/csrc/zkw_expand.hip -x c++ era-zk_evm_amd/csrc/zkw_pack.hip -x c++ era-zk_evm_amd/csrc/zkw_runtime.cpp -x c++ era-zk_evm_amd/csrc/isa_default.cpp -x c++ tests/emu/emu_glue.cpp -x c++ tests/emu/emu_simt.cpp"
COMMON="-O1 -g -std=c++17 -fPIC -shared -pthread -fno-omit-frame-pointer -Wno-unknown-pragmas -Wno-attributes -I tests/emu -I include"
echo "== one-lane emulation build, -fsanitize=address,undefined (-fno-sanitize-recover)"
g++ $COMMON -fsanitize=address,undefined -fno-sanitize-recover=undefined -DZKW_EMU_WAVE=1 -o /tmp/libzkw_emu1_asan.so $SRC || exit 1
EMULIB=/tmp/libzkw_emu1_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python profiles/tools/r10_sanitizer_run.py; echo "exit code $?"
echo "== 64-lane emulation build (SIMT engine), -fsanitize=undefined (-fno-sanitize-recover)"
g++ $COMMON -fsanitize=undefined -fno-sanitize-recover=undefined -DZKW_EMU_WAVE=64 -o /tmp/libzkw_emu64_ubsan.so $SRC || exit 1
EMULIB=/tmp/libzkw_emu64_ubsan.so LD_PRELOAD=$(gcc -print-file-name=libubsan.so) python profiles/tools/r10_sanitizer_run.py; echo "exit code $?"
echo "== one-lane emulation build, -fsanitize=thread (the delivery ring's worker threads replay beside the caller's thread)"
g++ $COMMON -fsanitize=thread -DZKW_EMU_WAVE=1 -o /tmp/libzkw_emu1_tsan.so $SRC || exit 1
EMULIB=/tmp/libzkw_emu1_tsan.so LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0" python profiles/tools/r10_sanitizer_run.py; echo "exit code $?"
