#!/bin/bash
# which part of the candidate walks breaks the 448x192 wavefront test with candidate slots?
mkdir -p gpurun_out/r03
T='tests/test_gpu_spine.py::test_encode_frame_wavefront_equals_the_reference_with_wavefront_synchro[448-192-3-5-16]'
run() { name=$1; shift; env "$@" timeout -k 10 200 python -m pytest "$T" -x -q > gpurun_out/r03/dbg_$name.log 2>&1; echo "$name: $(tail -n 1 gpurun_out/r03/dbg_$name.log)"; }
run nowalk HOP_WALK=0
run nointra HOP_WALK_NO_INTRA=1
run nointer HOP_WALK_NO_INTER=1
run nostreams HOP_SPINE_STREAMS=1
run default HOP_DUMMY=1
timeout -k 10 300 python -m pytest tests/test_gpu_tq_intra.py -x -q -k "device_classes" > gpurun_out/r03/t_walk1.log 2>&1; echo "device_classes: $(tail -n 1 gpurun_out/r03/t_walk1.log)"
