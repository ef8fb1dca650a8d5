cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, os, sys
sys.path.insert(0,'.')
from eesen_amd import kaldi_io, nnet_io, synth
rng=np.random.default_rng(5)
cfg=synth.config("tiny_bi")
feats=[(f"utt{i:02d}", rng.standard_normal((int(rng.integers(8,30)),cfg["D"])).astype(np.float32)) for i in range(14)]
feats.sort(key=lambda kv: kv[1].shape[0])
labs={k: rng.integers(1,cfg["K"],size=max(1,m.shape[0]//5)).astype(np.int32) for k,m in feats}
os.makedirs("/tmp/sd",exist_ok=True)
kaldi_io.write_mat_ark("/tmp/sd/feats.ark",feats,scp_path="/tmp/sd/feats.scp")
kaldi_io.write_vec_int_ark("/tmp/sd/labels.ark",labs.items())
nnet_io.write_nnet("/tmp/sd/nnet.init",synth.make_model(max_grad=50.0,**cfg),binary=True)
PY
which gdb valgrind catchsegv 2>&1 | head
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/eesen_amd/lib
gcc -shared -fPIC -o /tmp/segv_bt.so tools/segv_bt.c; if false; then
  gdb -batch -ex run -ex bt --args oracle/_ref/train-ctc-parallel-seam --learn-rate=0.01 --num-sequence=4 --frame-limit=90 scp:/tmp/sd/feats.scp ark:/tmp/sd/labels.ark /tmp/sd/nnet.init /tmp/sd/out.nnet 2>&1 | tail -40
else
  LD_PRELOAD=/tmp/segv_bt.so oracle/_ref/train-ctc-parallel-seam --learn-rate=0.01 --num-sequence=4 --frame-limit=90 scp:/tmp/sd/feats.scp ark:/tmp/sd/labels.ark /tmp/sd/nnet.init /tmp/sd/out.nnet; echo rc=$?
fi
