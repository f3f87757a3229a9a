#!/bin/bash
# round 4, v: is the flight of the attention kernel's LDS fragment reads a limit?  Ring depth 2 / 4 (shipped) / 8 with cycle counts per KV tile
# (python tools/attn_lab.py build --variants c_base,c_pf2,c_pf8)   -> gpurun_out/r4/v_attn_ring.txt
mkdir -p gpurun_out/r4
timeout 1500 python tools/attn_lab.py run --variants c_base,c_pf2,c_pf8 --rounds 2 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/v_attn_ring.txt
cat gpurun_out/r4/v_attn_ring.txt
