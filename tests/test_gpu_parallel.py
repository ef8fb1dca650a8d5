"""-m gpu: the RCCL plumbing of the data-parallel exchange on ONE GPU (world_size 1): the torch view aliases the
library's gradient buffer (zero copy), all_reduce runs on it in place and the update consumes what it left."""
import os
import socket

import numpy as np
import pytest

from eesen_amd import synth
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def test_grad_view_is_zero_copy_and_allreduce_in_place(gpu):
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from eesen_amd.api import Net, Ctc
    from eesen_amd.parallel import GradAllReducer, grad_tensor
    assert torch.cuda.is_available()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
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
        assert np.array_equal(net.GetGrads(), g0)
        t.mul_(2.0)                                # what two identical ranks would have summed to
        torch.cuda.synchronize()
        assert rel_err(net.GetGrads(), 2.0 * g0) < 1e-7
        p0 = net.GetParams()
        net.Update()
        assert rel_err(p0 - net.GetParams(), 2.0 * g0) < 1e-5
    finally:
        dist.destroy_process_group()
