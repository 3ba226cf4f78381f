#!/usr/bin/env python3
"""Freeze the reference's trained NN_11 parameters as DATA fixtures (run in the authoring container only).

  /root/reference/network/converged/Size_5_NN_11_17_Mar_2020_22_33_59.pt          -> nn11_d5_converged.safetensors
  /root/reference/network/converged/Size_7_NN_11_random_18_Mar_2020_18_17_52.pt   -> nn11_d7_converged.safetensors

The checkpoints are plain state_dicts (24 float32 tensors, ~0.90 M parameters; MIT-licensed upstream, LICENSE:1-3);
they are read with torch.load(..., weights_only=True) -- nothing in the file is executed -- and written back as
safetensors (tensors only, no pickle).  They are the "statistical oracle" of the env half: networks
trained against the real gym_ToricCode decode well only on an env with the same conventions, and the reference
records what they achieve (results/results_mats/RL_{5,7}.txt).  tests/test_gpu_accuracy.py drives the HIP env with
them; bench.py's nn_in_loop leg uses the d=7 set.
"""
import os
import sys

import torch
from safetensors.torch import save_file

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = {5: "network/converged/Size_5_NN_11_17_Mar_2020_22_33_59.pt",
       7: "network/converged/Size_7_NN_11_random_18_Mar_2020_18_17_52.pt"}

for d, rel in SRC.items():
    sd = torch.load(os.path.join(REF, rel), map_location="cpu", weights_only=True)
    assert all(torch.is_tensor(v) and v.dtype == torch.float32 for v in sd.values()) and len(sd) == 24
    out = os.path.join(HERE, f"nn11_d{d}_converged.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, out, metadata={"source": rel, "size": str(d)})
    print(out, sum(v.numel() for v in sd.values()), "parameters")
