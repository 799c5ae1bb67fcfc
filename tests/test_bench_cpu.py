"""bench.py's launcher contract without a GPU: `python bench.py --gpus N` must start N ranks by itself (it
exited with SystemExit unless torch.distributed.run had been used) and the rank count it reports is the one a collective
sees."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_n_ranks():
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--selftest-launch'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    last = [l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1]
    assert json.loads(last) == {'n_ranks': 2, 'world_size': 2}


def test_bench_rejects_a_launcher_rank_count_mismatch():
    env = dict(os.environ, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--selftest-launch'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b'launcher started 4 ranks' in p.stderr
