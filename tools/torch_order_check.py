"""Which of torch's and libhgs's HIP runtimes may come first?  (run on the GPU box; two subprocesses)"""
import subprocess
import sys

A = """
import os, sys
os.environ["HGS_SKIP_TORCH_INIT"] = "1"
sys.path.insert(0, ".")
import numpy as np
from slmsuite_amd.engine import Engine
from slmsuite_amd import _lib as L
e = Engine((256, 256), (64, 64)); e.set(L.PHASE, np.zeros((64, 64), np.float32)); e.nearfield2farfield()
import torch
print("engine first, then torch.cuda.is_available() ->", torch.cuda.is_available())
"""
B = """
import sys
sys.path.insert(0, ".")
import numpy as np
from slmsuite_amd.engine import Engine
from slmsuite_amd import _lib as L
e = Engine((256, 256), (64, 64)); e.set(L.PHASE, np.zeros((64, 64), np.float32)); e.nearfield2farfield()
import torch
print("default load order (torch runtime first) ->", torch.cuda.is_available(), torch.cuda.current_device())
t = torch.zeros((1, 64, 64), device="cuda")
e.get_into_device(L.PHASE, t.data_ptr(), 64 * 64 * 4)
print("device-to-torch copy ok", float(t.sum()))
"""
for code in (A, B):
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(p.stdout.strip(), "|", p.stderr.strip().splitlines()[-1] if p.returncode else "ok")
