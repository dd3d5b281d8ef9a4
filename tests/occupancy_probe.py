"""Diagnostic: chains per CU the runtime grants each build of the engine library (run on the GPU box)."""
import ctypes as C, os, sys, torch
torch.cuda.init()
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fast-cu-decision-hevc_amd")
p = torch.cuda.get_device_properties(0)
print("CUs", p.multi_processor_count)
for l in sys.argv[1:] or ["libfcu.so"]:
    lib = C.CDLL(os.path.join(here, l))
    print(l, "chains per CU:", lib.fcu_chains_per_cu())
