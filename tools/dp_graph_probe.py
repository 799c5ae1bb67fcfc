"""Probe (GPU box, 1 rank on RCCL, LSPS_FORCE_DP=1): which captured data-parallel update trips torch's RCCL watchdog?
usage: python tools/dp_graph_probe.py <dis|gen|post|all> [side|noside]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
os.environ['LSPS_FORCE_DP'] = '1'
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ['LSPS_BUCKET_BYTES'] = str(1 << 16)
what = sys.argv[1]
if len(sys.argv) > 2 and sys.argv[2] == 'noside':
    os.environ['LSPS_NO_OVERLAP'] = '1'
import torch
import torch.distributed as dist
import cases
from oracle import lsps_ref
import lsps_amd.trainers as prod

torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0), rank=0, world_size=1)
A = cases.NativeAdapter(prod, 'cuda')
hp = cases.hp_for('tiny')
sds = cases.make_weights(hp, lsps_ref)
lat2, lat1 = cases.latent_shape(hp, 8), cases.latent_shape(hp, 4)
zd = hp['vae']['z_dim']
tr = A.make_trainer(hp, sds)
tr.use_graphs(True)
A.set_train(tr, True)
b = cases.make_inputs(4)
for rnd in range(4):
    if what in ('dis', 'all'):
        A.dis_update(tr, b, hp, cases.noise(lat2, 10 + rnd))
    if what in ('gen', 'all'):
        A.gen_update(tr, b, hp, (cases.noise(lat2, 20 + rnd), cases.noise(lat1, 30 + rnd), cases.noise(lat1, 40 + rnd)))
    if what in ('post', 'all'):
        A.post_update(tr, b, 3, hp, cases.noise(lat2, 50 + rnd), cases.noise((4, zd), 60 + rnd, 0.05),
                      cases.noise((4, zd), 70 + rnd, 0.05))
    torch.cuda.synchronize()
    time.sleep(0.5)                       # give the watchdog thread time to poll whatever was enqueued
    print(what, 'round', rnd, 'ok, graphs', len(tr._graphs), flush=True)
dist.destroy_process_group()
print('PROBE_OK', what)
