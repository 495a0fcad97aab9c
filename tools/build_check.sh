#!/bin/bash
# rebuild the library; non-zero exit (and the compiler errors) on failure
out=$(python -m nunchaku_amd.build 2>&1); rc=$?
if [ $rc -ne 0 ]; then echo "$out" | grep -E "error" -A6 | head -30; echo "BUILD FAILED"; exit 1; fi
echo "build ok"
