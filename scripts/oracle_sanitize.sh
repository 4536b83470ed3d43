#!/bin/bash
# The CPU oracle under AddressSanitizer + UBSan (CPU only, no GPU): builds oracle/libssd_oracle_san.so and runs the oracle's own
# tests (known answers, properties, second opinion, parity-risk alternates) on it.  ASan must be the first DSO: LD_PRELOAD.
# usage: scripts/oracle_sanitize.sh [pytest args]      log: profiles/r05_oracle_sanitizers.log (written by the caller)
set -e
cd "$(dirname "$0")/.."
make -C oracle -s san
export SSD_ORACLE_LIB=$PWD/oracle/libssd_oracle_san.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1     # CPython itself leaks by design
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export OMP_NUM_THREADS=4
exec python -m pytest tests/test_oracle_known_answers.py tests/test_oracle_properties.py tests/test_oracle_second_opinion.py \
     tests/test_parity_risk_register.py tests/test_host.py -q -m "not gpu" -p no:cacheprovider "$@"
