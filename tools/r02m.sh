#!/bin/bash
# round 2, the last GPU seconds: the extension loop with the read window and the remembered splice site, 10 M fragments, budget 2048 (before: 122 + 453 ms)
mkdir -p gpurun_out
timeout 75 python tools/r02j.py 10000000 4096:21:2048 4096:21:2048 > gpurun_out/r02m_extension_ab.txt 2> gpurun_out/r02m_extension_ab.err
echo "exit $?"; cat gpurun_out/r02m_extension_ab.txt; tail -2 gpurun_out/r02m_extension_ab.err
