"""ctypes binding of libpamnet_hip.so (C ABI declared in include/pamnet_hip.h).

There is NO fallback: if the library is missing or a call returns non-zero, a RuntimeError is raised.  The product
path never imports oracle/ and never computes on the CPU.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
# (PAMNET_HIP_LIB: another build of the same ABI, for same-box A/B timing of kernel changes -- a developer switch)
LIB_PATH = os.environ.get('PAMNET_HIP_LIB') or os.path.join(HERE, 'libpamnet_hip.so')
HEADER = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'pamnet_hip.h')

_lib = None
_CT = {'float*': ctypes.c_void_p, 'int32_t*': ctypes.c_void_p, 'void*': ctypes.c_void_p,
       'int64_t': ctypes.c_int64, 'int32_t': ctypes.c_int32, 'float': ctypes.c_float, 'double': ctypes.c_double,
       'pamnet_stream_t': ctypes.c_void_p, 'int': ctypes.c_int}


def declared_functions(header=HEADER):
    """Parse `int pamnet_xxx(args);` prototypes out of the public header -> {name: [ctypes arg types]}."""
    text = open(header).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    out = {}
    for m in re.finditer(r'\bint\s+(pamnet_\w+)\s*\(([^)]*)\)\s*;', text):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.replace('const', ' ').strip()
                if '*' in a:                      # every pointer (device or host array) travels as void*
                    types.append(ctypes.c_void_p)
                else:
                    types.append(_CT[a.rsplit(None, 1)[0].strip()])
        out[name] = types
    return out


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libpamnet_hip.so is not built (%s). Run `python __graft_entry__.py` or '
                               'pamnet_amd.build.build(); there is no CPU fallback.' % LIB_PATH)
        import torch  # noqa: F401  -- FIRST: the library must bind to the HIP runtime torch ships (one runtime per
        #                              process; loading /opt/rocm's copy beside torch's yields hipErrorNoDevice)
        lib = ctypes.CDLL(LIB_PATH)
        for name, types in declared_functions().items():
            fn = getattr(lib, name)            # AttributeError here = header/library mismatch: fail loudly
            fn.argtypes = types
            fn.restype = ctypes.c_int
        want = int(re.search(r'#define\s+PAMNET_ABI_VERSION\s+(\d+)', open(HEADER).read()).group(1))
        lib.pamnet_abi_version.argtypes = []
        got = int(lib.pamnet_abi_version())
        if got != want:
            raise RuntimeError('libpamnet_hip.so was built for ABI version %d, include/pamnet_hip.h declares %d: rebuild '
                               '(python __graft_entry__.py)' % (got, want))
        if os.environ.get('PAMNET_HIP_LIB'):
            # a developer switch must never be silent: the ABI number does not tell two builds of the same ABI apart
            import warnings
            warnings.warn('pamnet_amd: PAMNET_HIP_LIB is set -- kernels come from %s (ABI %d), not from the in-tree build %s'
                          % (LIB_PATH, got, os.path.join(HERE, 'libpamnet_hip.so')), RuntimeWarning, stacklevel=2)
        _lib = lib
    return _lib


def check(rc, name):
    if rc != 0:
        kind = {-1: 'PAMNET_EINVAL (bad size / unsupported width)', -2: 'PAMNET_ENULL (null pointer)'}.get(
            rc, 'hipError_t %d' % rc)
        raise RuntimeError('%s failed: %s' % (name, kind))


_EMPTY = {}


def ptr(t):
    """Device pointer of a contiguous tensor, or NULL for None.  A zero-length tensor has no storage (data_ptr() == 0),
    which the library would report as a missing argument: it gets the address of a small per-device placeholder that is
    never dereferenced (every kernel is bounded by the row count, which is 0)."""
    if t is None:
        return None
    p = t.data_ptr()                               # (called ~200 times per step on the host-bound narrow-width paths)
    if not t.is_contiguous():
        raise AssertionError('pamnet_hip: tensor must be contiguous')
    if p == 0 and t.is_cuda:
        import torch
        if t.device not in _EMPTY:
            _EMPTY[t.device] = torch.zeros(64, dtype=torch.float32, device=t.device)
        return _EMPTY[t.device].data_ptr()
    return p


_raw_stream = None


def stream_of(t):
    """hipStream_t of torch's current stream on the tensor's device.  (torch.cuda.current_stream() builds a Stream object
    per call: ~5 us, 50-300 times per step on the per-operator paths; the raw accessor is a plain C call.)"""
    global _raw_stream
    if not t.is_cuda:
        raise RuntimeError('pamnet_hip kernels run on an MI355X only: tensor is on %s (no CPU fallback)' % t.device)
    if _raw_stream is None:
        import torch
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or \
            (lambda idx: torch.cuda.current_stream(idx).cuda_stream)
    return _raw_stream(t.device.index)


_fns = {}


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    rc = fn(*args)
    if rc:
        check(rc, name)


_P = ctypes.c_void_p


class EmbedJob(ctypes.Structure):
    """pamnet_embed_job (include/pamnet_hip.h)."""
    _fields_ = [('x', _P), ('dist', _P), ('freq', _P), ('cutoff', ctypes.c_float), ('K', ctypes.c_int32),
                ('act', ctypes.c_int32), ('rows', ctypes.c_int64), ('kind', _P), ('W0', _P), ('b0', _P), ('W1', _P),
                ('b1', _P), ('out', _P), ('gout', _P), ('dW0', _P), ('db0', _P), ('dW1', _P), ('db1', _P), ('dfreq', _P),
                ('dx', _P), ('partial', _P)]


class TypeRowsJob(ctypes.Structure):
    """pamnet_type_rows_job (include/pamnet_hip.h)."""
    _fields_ = [('table', _P), ('idx', _P), ('n', ctypes.c_int64), ('n_types', ctypes.c_int64), ('out', _P), ('g', _P),
                ('scratch', _P), ('dtable', _P)]


class GraphDesc(ctypes.Structure):
    """pamnet_graph_desc (include/pamnet_hip.h): one batch handed to the graph-construction engine."""
    _fields_ = [('n', ctypes.c_int64), ('n_graphs', ctypes.c_int64), ('n_bonds', ctypes.c_int64),
                ('eg', ctypes.c_int64), ('el', ctypes.c_int64), ('tp', ctypes.c_int64),
                ('batch', _P), ('types', _P), ('types_stride', ctypes.c_int64), ('n_types', ctypes.c_int64),
                ('pos', _P), ('rows', _P), ('rows_width', ctypes.c_int64), ('edge_src', _P), ('edge_dst', _P),
                ('schema', ctypes.c_int32), ('batch_kind', ctypes.c_int32), ('types_kind', ctypes.c_int32),
                ('edge_kind', ctypes.c_int32), ('with_triplets', ctypes.c_int32), ('need_grad', ctypes.c_int32),
                ('aggregate_at_query', ctypes.c_int32), ('knn_k', ctypes.c_int32),
                ('cutoff_l', ctypes.c_float), ('cutoff_g', ctypes.c_float), ('max_neighbors', ctypes.c_int32),
                ('mol_local', ctypes.c_int32)]


class MolGraphOut(ctypes.Structure):
    """pamnet_mol_graph_out (include/pamnet_hip.h): output arrays of the molecule-local graph builder."""
    _fields_ = [(k, _P) for k in ('g_ptr', 'g_row', 'g_col', 'g_dist', 'gT_perm', 'l_ptr', 'l_row', 'l_col', 'l_dist',
                                  'lT_ptr', 'lT_perm', 't_ptr', 't_row', 't_col', 't_angle', 't_kind', 'tT_ptr', 'tT_perm')]


# field indices of the engine's arena layout (enum PAMNET_GF_* in include/pamnet_hip.h, same order)
GF = {name: i for i, name in enumerate(
    ['NODE_GRAPH', 'GPTR', 'FLAG', 'LOOPS', 'TYPES', 'POS', 'SIGN', 'G_PTR', 'G_ROW', 'G_COL', 'G_DIST', 'GT_PTR', 'GT_PERM',
     'L_PTR', 'L_ROW', 'L_COL', 'L_DIST', 'LT_PTR', 'LT_PERM', 'T_PTR', 'T_ROW', 'T_COL', 'T_ANGLE', 'T_KIND', 'TT_PTR',
     'TT_PERM', 'CUTS', 'TT_EDGE', 'TT_NODE'])}
GRAPH_FIELDS = len(GF)
