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
    assert e.side_prio == -1 and e.share_encoder and e.wino == 3
    # the library's own switches are ordinary fields now (round 6: no getenv in csrc); native() is the lsps_set_options block
    assert d.native() == dict(wino4_split=1, fs2_cc=4, wino4w=1, wino4w_waves=8, chwn_group=1, c8w_queue=1, c8_stem_bf16=1, x3_plan=1, x3_ring=0)
    n = options.from_env({'LSPS_WINO4_SPLIT': '0', 'LSPS_FS2_CC': '8', 'LSPS_WINO4W': '0', 'LSPS_WINO4W_WAVES': '4', 'LSPS_CHWN_GROUP': '0',
                          'LSPS_C8W_QUEUE': '2', 'LSPS_C8_STEM_BF16': '0', 'LSPS_X3_PLAN': '0', 'LSPS_X3_RING': '1', 'LSPS_HIP_LIB': '/x/y.so'})
    assert n.native() == dict(wino4_split=0, fs2_cc=8, wino4w=0, wino4w_waves=4, chwn_group=0, c8w_queue=2, c8_stem_bf16=0, x3_plan=0, x3_ring=1)
    assert n.hip_lib == '/x/y.so' and hash(n) != hash(d)
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


def test_the_library_reads_no_environment_variable():
    """every dispatch switch inside liblsps_hip.so comes through lsps_set_options."""
    for root, _, files in os.walk(os.path.join(REPO, 'lsps_amd', 'csrc')):
        for f in files:
            if f.endswith(('.hip', '.h')):
                assert 'getenv' not in open(os.path.join(root, f)).read(), f
    hdr = open(os.path.join(REPO, 'include', 'lsps_hip.h')).read()
    from lsps_amd import options, _lib
    body = hdr[hdr.index('typedef struct LspsOptions {'):hdr.index('} LspsOptions;')]
    fields = re.findall(r'^\s*int\s+(\w+);', body, re.M)
    assert fields == ['struct_size'] + list(options.NATIVE_FIELDS) == [f for f, _ in _lib.LspsOptions._fields_]


def test_set_options_round_trips_through_the_library_and_follows_the_options_object():
    from lsps_amd import options, _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('library not built')
    before = options.get()
    assert _lib.native_options() == before.native()                  # pushed when the library was loaded
    with options.override(fs2_cc=8, wino4w=False, x3_plan=0, c8w_queue=2):
        got = _lib.native_options()
        assert (got['fs2_cc'], got['wino4w'], got['x3_plan'], got['c8w_queue']) == (8, 0, 0, 2)
    assert _lib.native_options() == before.native()                  # and restored with the object
    bad = _lib.LspsOptions(struct_size=4)
    assert _lib.lib().lsps_set_options(bad) != 0 and b'struct_size' in _lib.lib().lsps_last_error()
    blk = _lib.LspsOptions(struct_size=__import__('ctypes').sizeof(_lib.LspsOptions), **dict(before.native(), fs2_cc=5))
    assert _lib.lib().lsps_set_options(blk) != 0                     # rejected, nothing changed
    assert _lib.native_options() == before.native()


def test_warns_when_lsps_variables_change_after_import(monkeypatch):
    from lsps_amd import options
    assert options.warn_if_env_changed() == []
    monkeypatch.setenv('LSPS_X3', '0')
    with pytest.warns(RuntimeWarning, match='LSPS_X3'):
        assert options.warn_if_env_changed() == ['LSPS_X3']
    prev = options.get()
    try:
        assert not options.reload_env().x3 and options.warn_if_env_changed() == []
    finally:
        monkeypatch.delenv('LSPS_X3')
        options.reload_env()
        assert options.get() == prev
