#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_vae.py::test_upsample_conv_as_four_phase_convs tests/test_gpu_bsa.py::test_fused_topk_edge_shapes -q -x 2>&1 | tail -15 > gpurun_out/edge.log
cat gpurun_out/edge.log
