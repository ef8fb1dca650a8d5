"""Run as a subprocess by tests/test_gpu_parallel.py (torch must be imported BEFORE libeesen_hip.so is loaded so that
both share one HIP runtime instance — torch ships its own libamdhip64.so.7)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert torch.cuda.is_available(), "torch sees no GPU"
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

from eesen_amd import synth  # noqa: E402
from eesen_amd.api import Net, Ctc  # noqa: E402
from eesen_amd.parallel import GradAllReducer, grad_tensor  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(float(np.max(np.abs(b))), 1e-12))


cfg = synth.config("small_bi")
layers = synth.make_model(**cfg)
batch = synth.make_batch(**cfg)
net = Net.from_layers(layers, stream=torch.cuda.current_stream().cuda_stream or None)
net.SetTrainOptions(1.0, 0.0)
ctc = Ctc()
net.SetSeqLengths(batch.lens)
out = net.Propagate(batch.feats)
diff = ctc.EvalParallel(batch.lens, out, batch.labels)
net.BackpropagateNoUpdate(diff)
g0 = net.GetGrads()
t = grad_tensor(net)
ptr, n = net.grad_buffer()
assert t.data_ptr() == ptr and t.numel() == n
red = GradAllReducer(net)
red(net)                                   # world size 1: the sum over ranks is the identity
torch.cuda.synchronize()
assert np.array_equal(net.GetGrads(), g0), "all_reduce over one rank changed the gradients"
t.mul_(2.0)                                # what two identical ranks would have summed to
torch.cuda.synchronize()
assert rel(net.GetGrads(), 2.0 * g0) < 1e-7, "the torch view does not alias the library's gradient buffer"
p0 = net.GetParams()
net.Update()
assert rel(p0 - net.GetParams(), 2.0 * g0) < 1e-5
dist.destroy_process_group()
print("RCCL_PLUMBING_OK")
