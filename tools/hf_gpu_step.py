"""The reference's model class on the same GPU (bench.py reference_class_on_gpu): transformers.VideoMAEForPreTraining at the bench workload with PyTorch-ROCm's
own attention ("sdpa") and with only the attention swapped for the HIP kernels through the attention registry ("smbv").  usage: hf_gpu_step.py [steps]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.reference_class_on_gpu(torch.device("cuda", 0), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 5), indent=1))
