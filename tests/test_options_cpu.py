"""The process options object (lsps_amd/options.py): every LSPS_* switch of the product path read once into a frozen,
hashable dataclass; nothing under lsps_amd/trainers/ reads the environment for dispatch any more."""
import json
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_from_env_parses_every_switch_and_is_hashable():
    from lsps_amd import options
    d = options.from_env({})
    assert d.chwn and d.overlap and d.pack_cache and d.frozen_packs and d.est_merge and d.x3 and d.fuse_act and not d.force_dp
    assert d.chwn_min_n == 96 and d.x3_min_gmac == 1.0 and d.bucket_bytes == options.DEFAULT_BUCKET_BYTES and not d.share_encoder
    e = options.from_env({'LSPS_CHWN': '0', 'LSPS_CHWN_MIN_N': '16', 'LSPS_NO_OVERLAP': '1', 'LSPS_NO_PACK_CACHE': '1',
                          'LSPS_NO_FROZEN_PACKS': '1', 'LSPS_EST_SPLIT_BACKWARD': '0', 'LSPS_EST_ORDER': 'chain', 'LSPS_EST_MERGE': '0',
                          'LSPS_FUSE_ACT': '0', 'LSPS_C8_FUSE_ACT': '0', 'LSPS_C8': '0', 'LSPS_C8S2': '0', 'LSPS_X3': '0',
                          'LSPS_X3_MIN_GMAC': '7.5', 'LSPS_FORCE_DP': '1', 'LSPS_DP_GRAPHS': '0', 'LSPS_BUCKET_BYTES': '65536',
                          'LSPS_SIDE_PRIO': '-1', 'LSPS_SHARE_ENCODER': '1', 'LSPS_WINO': '3'})
    assert not (e.chwn or e.overlap or e.pack_cache or e.frozen_packs or e.est_split_backward or e.est_merge or e.fuse_act or e.c8_fuse_act
                or e.c8 or e.c8s2 or e.x3 or e.dp_graphs)
    assert e.chwn_min_n == 16 and e.est_order == 'chain' and e.x3_min_gmac == 7.5 and e.force_dp and e.bucket_bytes == 65536
    assert e.side_prio == -1 and e.share_encoder and dict(e.native) == {'LSPS_WINO': '3'}
    assert hash(d) != hash(e) and d != e and d == options.from_env({})           # usable inside a hipGraph signature
    json.dumps(e.as_dict())                                                       # goes into bench.py's JSON line
    with pytest.raises(Exception):
        d.x3 = False                                                              # frozen


def test_override_restores_and_set_replaces():
    from lsps_amd import options
    before = options.get()
    with options.override(x3=False, chwn_min_n=4) as o:
        assert options.get() is o and not o.x3 and o.chwn_min_n == 4
    assert options.get() is before
    prev = options.set(overlap=False)
    try:
        assert prev is before and not options.get().overlap
    finally:
        options.restore(prev)
    assert options.get() is before


def test_no_environment_reads_left_in_the_trainers_package():
    pat = re.compile(r"environ|getenv")
    for root, _, files in os.walk(os.path.join(REPO, 'lsps_amd', 'trainers')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert not pat.search(src), os.path.join(root, f)
    # the rest of the product path reads LSPS_* only in options.py (and the library path in _lib.py)
    for f in ('ops.py', 'dist.py', 'optim.py'):
        src = open(os.path.join(REPO, 'lsps_amd', f)).read()
        assert 'LSPS_' not in ''.join(l for l in src.splitlines(True) if 'environ' in l), f
