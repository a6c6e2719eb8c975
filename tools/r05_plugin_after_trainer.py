"""Diagnostic (round 5): bench.py's reference_class_on_gpu leg read 70.7 ms for 'smbv' inside the full bench and 47.2 ms alone / without the Trainer leg.  This runs the
Trainer leg first, then the plugin leg, and prints torch's global switches before / after, so that rocprofv3 --stats over it shows what got slower."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def switches():
    return dict(threads=torch.get_num_threads(), det=torch.are_deterministic_algorithms_enabled(), fill=torch.utils.deterministic.fill_uninitialized_memory,
                autocast=torch.is_autocast_enabled(), anomaly=torch.is_anomaly_enabled(), grad=torch.is_grad_enabled(), bench=torch.backends.cudnn.benchmark,
                matmul_prec=torch.get_float32_matmul_precision(), sdp_flash=torch.backends.cuda.flash_sdp_enabled(), sdp_mem=torch.backends.cuda.mem_efficient_sdp_enabled(),
                sdp_math=torch.backends.cuda.math_sdp_enabled(), default_dtype=str(torch.get_default_dtype()))

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
print("before", switches(), file=sys.stderr)
if len(sys.argv) > 1 and sys.argv[1] == "trainer":
    print("trainer ms", bench.trainer_ms_per_step(dict(bench.VIT_BASE), steps=4, warm=4), file=sys.stderr)
print("after", switches(), file=sys.stderr)
print(json.dumps(bench.reference_class_on_gpu(dev, steps=3)))
