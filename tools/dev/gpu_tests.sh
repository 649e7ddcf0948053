#!/bin/bash
# GPU job: the GPU test suite (or the files given), full failure output kept in gpurun_out/pytest.log
mkdir -p gpurun_out
python -m pytest ${@:-tests} -m gpu -q -x 2>&1 | tail -80 > gpurun_out/pytest.log
tail -5 gpurun_out/pytest.log
