"""bench.py's roofline leg alone (the decoder attention backward, timed with HIP events on its launch stream): prints {"pair": ms, "dkv": ms, "dq": ms}."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
res, unit = bench.kernel_roofline(dev)
print(json.dumps({k: round(v, 4) for k, v in res.items()}))
