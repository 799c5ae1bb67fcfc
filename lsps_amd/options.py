"""Process options of the product path: every `LSPS_*` switch, read ONCE (at import) into one frozen object.

Dispatch decided by scattered `os.environ.get` calls is hard to reason about under eight ranks and the hipGraph
signature had to enumerate the switches by hand.  Now:

  * `options.get()` is the object in force — a frozen dataclass, hashable: `LSPSTrainer._graphed` puts it into the graph
    signature as a whole, so a switch added later cannot be forgotten there;
  * `options.set(**kw)` / `with options.override(**kw):` replace it programmatically (tests, tools, the bench's A/B legs);
  * `options.get().as_dict()` goes into the bench's JSON line, so a multi-rank run records what it ran with;
  * the switches that act INSIDE the shared library (`wino4_split` ... `x3_plan`, and `wino` = the initial Winograd mode) are
    fields like any other: the library reads no environment variable (round 6); `_push()` hands them over through the one
    C-ABI call `lsps_set_options` (include/lsps_hip.h) when the library is loaded (`_lib.lib()`) and again whenever the
    object in force changes, so the graph signature and the bench line describe what the kernels really ran with.
  * entry points (`LSPSTrainer.__init__`, `dist.init`) call `warn_if_env_changed()`: a launcher that sets `LSPS_*` after this
    module was imported is told that the value is ignored until `reload_env()`.
"""
import contextlib
import dataclasses
import os

DEFAULT_BUCKET_BYTES = 16 << 20

# fields handed to the library by lsps_set_options (include/lsps_hip.h: struct LspsOptions, same order after struct_size)
NATIVE_FIELDS = ('wino4_split', 'fs2_cc', 'wino4w', 'wino4w_waves', 'chwn_group', 'c8w_queue', 'c8_stem_bf16', 'x3_plan', 'x3_ring')


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
    c8_fuse_act: bool = True           # LSPS_C8_FUSE_ACT=0: the same, C8 and three-limb (X3) kernels only
    c8: bool = True                    # LSPS_C8=0: bf16 mode without the C8 layout
    c8s2: bool = True                  # LSPS_C8S2=0: bf16 mode without the C8 stride-2 family
    x3: bool = True                    # LSPS_X3=0: f32 mode without the three-limb stride-2 family (csrc/x3s2.h)
    x3_min_gmac: float = 1.0           # LSPS_X3_MIN_GMAC: smallest layer (10^9 multiply-adds per launch) routed to it
    force_dp: bool = False             # LSPS_FORCE_DP=1: gradient exchange also in a 1-rank group
    dp_graphs: bool = True             # LSPS_DP_GRAPHS=0: data-parallel steps never captured
    bucket_bytes: int = DEFAULT_BUCKET_BYTES   # LSPS_BUCKET_BYTES
    lazy_scalars: bool = True          # LSPS_LAZY_SCALARS=0: every update method ends with a synchronous device -> host copy of its scalars
    # ---- inside the library (csrc), through lsps_set_options ----
    wino: int = 1                      # LSPS_WINO: initial Winograd mode 0..4 (lsps_set_winograd; ops.set_winograd changes it later)
    wino4_split: bool = True           # LSPS_WINO4_SPLIT=0: no reduction-split F(4x4,3x3) launches
    fs2_cc: int = 4                    # LSPS_FS2_CC=4|8: channel chunk of the exact-f32 3x3 / stride-2 forward kernel
    wino4w: bool = True                # LSPS_WINO4W=0: weight gradient without the F(4x4,3x3) kernel
    wino4w_waves: int = 8              # LSPS_WINO4W_WAVES=4|8
    chwn_group: bool = True            # LSPS_CHWN_GROUP=0: trunk dgrad with one workgroup set per position
    c8w_queue: int = 1                 # LSPS_C8W_QUEUE: workgroups per CU of the C8 weight-gradient grids
    c8_stem_bf16: bool = True          # LSPS_C8_STEM_BF16=0: f32 stems in bf16 mode
    x3_plan: int = 1                   # LSPS_X3_PLAN=0: three-limb kernels launched without a plan
    x3_ring: bool = False              # LSPS_X3_RING=1: three-limb forward kernel with the 3-deep image ring (round 6 experiment)
    hip_lib: str = ''                  # LSPS_HIP_LIB: another build of the library (kernel A/B experiments); '' = in-tree

    def as_dict(self):
        return dataclasses.asdict(self)

    def native(self):
        """The block lsps_set_options receives: {field: int}."""
        return dict((k, int(getattr(self, k))) for k in NATIVE_FIELDS)


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
        bucket_bytes=int(e.get('LSPS_BUCKET_BYTES', DEFAULT_BUCKET_BYTES)), lazy_scalars=off('LSPS_LAZY_SCALARS'),
        wino=int(e.get('LSPS_WINO', '1')), wino4_split=off('LSPS_WINO4_SPLIT'), fs2_cc=int(e.get('LSPS_FS2_CC', '4')),
        wino4w=off('LSPS_WINO4W'), wino4w_waves=int(e.get('LSPS_WINO4W_WAVES', '8')), chwn_group=off('LSPS_CHWN_GROUP'),
        c8w_queue=int(e.get('LSPS_C8W_QUEUE', '1')), c8_stem_bf16=off('LSPS_C8_STEM_BF16'), x3_plan=int(e.get('LSPS_X3_PLAN', '1')), x3_ring=on('LSPS_X3_RING'),
        hip_lib=e.get('LSPS_HIP_LIB', ''))


def _env_snapshot():
    return dict((k, v) for k, v in os.environ.items() if k.startswith('LSPS_'))


_current = from_env()
_env_seen = _env_snapshot()


def _push():
    """Hands the library's share of the options in force to the library, if it is loaded (`_lib.lib()` calls this on load)."""
    from . import _lib
    if _lib._lib is not None:
        _lib.push_options(_current)


def warn_if_env_changed():
    """Entry points call this: `LSPS_*` variables set AFTER the import of this module are not in force."""
    now = _env_snapshot()
    if now != _env_seen:
        import warnings
        diff = sorted(k for k in dict(_env_seen, **now) if now.get(k) != _env_seen.get(k))   # (`set` is this module's setter)
        warnings.warn("lsps_amd.options: %s changed after the options were read at import; call lsps_amd.options.reload_env() "
                      "(or options.set(...)) for the new values to take effect" % ', '.join(diff), RuntimeWarning, stacklevel=2)
        return diff
    return []


def get():
    return _current


def set(**kw):
    """Replaces fields of the options in force; returns the previous object (pass it to `restore`)."""
    global _current
    prev = _current
    _current = dataclasses.replace(_current, **kw)
    if _current.native() != prev.native():
        _push()
    return prev


def restore(prev):
    global _current
    changed = _current.native() != prev.native()
    _current = prev
    if changed:
        _push()


def reload_env():
    """Re-reads the environment (worker processes that set `LSPS_*` after this module was imported by their parent)."""
    global _current, _env_seen
    _current = from_env()
    _env_seen = _env_snapshot()
    _push()
    return _current


@contextlib.contextmanager
def override(**kw):
    prev = set(**kw)
    try:
        yield _current
    finally:
        restore(prev)
