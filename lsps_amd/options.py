"""Process options of the product path: every `LSPS_*` switch, read ONCE (at import) into one frozen object.

Dispatch decided by scattered `os.environ.get` calls is hard to reason about under eight ranks and the hipGraph
signature had to enumerate the switches by hand.  Now:

  * `options.get()` is the object in force — a frozen dataclass, hashable: `LSPSTrainer._graphed` puts it into the graph
    signature as a whole, so a switch added later cannot be forgotten there;
  * `options.set(**kw)` / `with options.override(**kw):` replace it programmatically (tests, tools, the bench's A/B legs);
  * `options.get().as_dict()` goes into the bench's JSON line, so a multi-rank run records what it ran with;
  * `native` lists the switches the shared library reads itself (csrc: `getenv` once per process, cached in statics) —
    recorded here for the same reason, not interpreted.
"""
import contextlib
import dataclasses
import os

DEFAULT_BUCKET_BYTES = 16 << 20

# switches the library reads (csrc/igemm.hip, chwn.hip, c8.hip); values are fixed for the life of the process
_NATIVE = ('LSPS_WINO', 'LSPS_WINO4_SPLIT', 'LSPS_FS2_CC', 'LSPS_WINO4W', 'LSPS_WINO4W_WAVES', 'LSPS_CHWN_GROUP', 'LSPS_C8W_QUEUE',
           'LSPS_C8_STEM_BF16', 'LSPS_X3_PLAN', 'LSPS_HIP_LIB')


@dataclasses.dataclass(frozen=True)
class Options:
    chwn: bool = True                  # LSPS_CHWN=0: discriminator trunk stays NCHW (csrc/chwn.hip off)
    chwn_min_n: int = 96               # LSPS_CHWN_MIN_N: smallest batch that takes the batch-innermost trunk
    overlap: bool = True               # LSPS_NO_OVERLAP=1: estimate modes on one stream
    side_prio: int = 0                 # LSPS_SIDE_PRIO: priority of the side stream
    pack_cache: bool = True            # LSPS_NO_PACK_CACHE=1: pack weights per call
    frozen_packs: bool = True          # LSPS_NO_FROZEN_PACKS=1: generator panels re-packed per estimate step
    est_split_backward: bool = True    # LSPS_EST_SPLIT_BACKWARD=0: one backward over the summed estimate loss
    est_order: str = 'feat_first'      # LSPS_EST_ORDER: feat_first | reg_first | chain
    share_encoder: bool = False        # LSPS_SHARE_ENCODER=1: gen_update reuses the encoder pass of the dis_update in front of it
    est_merge: bool = True             # LSPS_EST_MERGE=0: estimate modes run dis.regress_* and dis.feats as two passes (round 4)
    fuse_act: bool = True              # LSPS_FUSE_ACT=0: LeakyReLU backward as separate passes (f32 and C8)
    c8_fuse_act: bool = True           # LSPS_C8_FUSE_ACT=0: the same, C8 kernels only
    c8: bool = True                    # LSPS_C8=0: bf16 mode without the C8 layout
    c8s2: bool = True                  # LSPS_C8S2=0: bf16 mode without the C8 stride-2 family
    x3: bool = True                    # LSPS_X3=0: f32 mode without the three-limb stride-2 family (csrc/x3s2.h)
    x3_min_gmac: float = 1.0           # LSPS_X3_MIN_GMAC: smallest layer (10^9 multiply-adds per launch) routed to it
    force_dp: bool = False             # LSPS_FORCE_DP=1: gradient exchange also in a 1-rank group
    dp_graphs: bool = True             # LSPS_DP_GRAPHS=0: data-parallel steps never captured
    bucket_bytes: int = DEFAULT_BUCKET_BYTES   # LSPS_BUCKET_BYTES
    native: tuple = ()                 # ((name, value), ...) of the library's own switches that are set

    def as_dict(self):
        d = dataclasses.asdict(self)
        d['native'] = dict(self.native)
        return d


def from_env(env=None):
    e = os.environ if env is None else env

    def off(name):                      # default on, "0" switches off
        return e.get(name, '1') != '0'

    def on(name):                       # default off, "1" switches on
        return e.get(name) == '1'
    return Options(
        chwn=off('LSPS_CHWN'), chwn_min_n=int(e.get('LSPS_CHWN_MIN_N', '96')),
        overlap=not on('LSPS_NO_OVERLAP'),
        side_prio=int(e.get('LSPS_SIDE_PRIO', '0')), pack_cache=not on('LSPS_NO_PACK_CACHE'),
        frozen_packs=not on('LSPS_NO_FROZEN_PACKS'), est_split_backward=off('LSPS_EST_SPLIT_BACKWARD'),
        est_order=e.get('LSPS_EST_ORDER', 'feat_first'), est_merge=off('LSPS_EST_MERGE'), share_encoder=on('LSPS_SHARE_ENCODER'), fuse_act=off('LSPS_FUSE_ACT'), c8_fuse_act=off('LSPS_C8_FUSE_ACT'),
        c8=off('LSPS_C8'), c8s2=off('LSPS_C8S2'), x3=off('LSPS_X3'), x3_min_gmac=float(e.get('LSPS_X3_MIN_GMAC', '1.0')), force_dp=on('LSPS_FORCE_DP'), dp_graphs=off('LSPS_DP_GRAPHS'),
        bucket_bytes=int(e.get('LSPS_BUCKET_BYTES', DEFAULT_BUCKET_BYTES)),
        native=tuple((k, e[k]) for k in _NATIVE if k in e))


_current = from_env()


def get():
    return _current


def set(**kw):
    """Replaces fields of the options in force; returns the previous object (pass it to `restore`)."""
    global _current
    prev = _current
    _current = dataclasses.replace(_current, **kw)
    return prev


def restore(prev):
    global _current
    _current = prev


def reload_env():
    """Re-reads the environment (worker processes that set `LSPS_*` after this module was imported by their parent)."""
    global _current
    _current = from_env()
    return _current


@contextlib.contextmanager
def override(**kw):
    prev = set(**kw)
    try:
        yield _current
    finally:
        restore(prev)
