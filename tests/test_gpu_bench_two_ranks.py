"""-m gpu: bench.py's multi-rank branch (init_process_group, rank-seeded data, the overlapped two-bucket exchange, barriers, max over
ranks, dp_check) as TWO processes sharing the one GPU of a test box — exactly the command line the driver uses for N = 2, with
OCR_DIST_BACKEND=gloo as the transport (RCCL needs one GPU per rank).  VERDICT r2: that branch had never executed anywhere."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("launcher", [True, False])
def test_bench_line_of_two_ranks(dev, launcher):
    """launcher=False: the bare `python bench.py --gpus 2` form — bench.py re-executes itself under torch.distributed.run (VERDICT r3 item 7:
    that form used to exit without a line)."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, OCR_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    tail = [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '3']
    cmd = ([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
            '--master-port', str(port)] if launcher else [sys.executable]) + tail
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['warmup'] == 3 and d['config']['global_batch'] == 128
    assert d['config']['parallelism'] == 'dp2' and d['scaling'] == 'weak' and d['value'] > 0
    chk = d['dp_check']
    assert chk['replicas_bit_identical'] is True                    # every rank applied the same exchanged gradient
    assert chk['local_loss_max'] > chk['local_loss_min']            # rank-seeded data streams: the local losses differ
    assert 'cpu_baseline' not in d                                  # rank 0 at N = 1 only
    assert d['dp_host_enqueue_us'] and d['dp_host_enqueue_us']['graph1_fwd_ctc_bwd_late'] > 0
