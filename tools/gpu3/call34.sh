#!/bin/bash
# whole -m gpu suite + smoke() on the current code
mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r03/suite3_tests.log 2>&1
rc=$?
tail -6 gpurun_out/r03/suite3_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/suite3_smoke.log 2>&1
echo "smoke rc=$? $(tail -1 gpurun_out/r03/suite3_smoke.log)"
exit $rc
