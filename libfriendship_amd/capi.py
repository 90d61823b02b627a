"""ctypes binding of include/friendship_render.h.

The same binding drives the product library (libfriendship_hip.so, HIP/gfx950) and -- from tests and
the bench's cpu_baseline leg only -- the CPU oracle (oracle/_build/libfr_oracle.so): both export the
identical C ABI.  This module contains no compute and no fallback: it marshals arguments.
"""
import ctypes as C
import json
import os

import numpy as np

# --- enums (friendship_render.h) ---------------------------------------------------------------
FR_OK = 0
FR_ERR_INVALID_ARG = 1
FR_ERR_INPUT_TOO_LONG = 2
FR_ERR_INPUT_HISTORY = 3
FR_ERR_NO_SUCH_NODE = 4
FR_ERR_BAD_SLOT = 5
FR_ERR_CYCLE = 6
FR_ERR_DEVICE = 7
FR_ERR_NO_DEVICE = 8
FR_ERR_OUT_OF_MEMORY = 9
FR_ERR_UNSUPPORTED = 10
FR_ERR_COMM = 11

# routing::effect::PrimitiveEffect declaration order (reference src/routing/effect.rs:86-112)
PRIMITIVES = ("Delay", "F32Constant", "Sum2", "Multiply", "Divide", "Modulo", "Minimum")
FR_PRIM = {name: i for i, name in enumerate(PRIMITIVES)}
FR_EFFECT_GRAPH = 7

FR_MODE_AUTO, FR_MODE_PULL, FR_MODE_STAGED = 0, 1, 2
MODES = {"auto": FR_MODE_AUTO, "pull": FR_MODE_PULL, "staged": FR_MODE_STAGED}
FR_SEMANTICS_REFERENCE, FR_SEMANTICS_SPARKLE = 0, 1
SEMANTICS = {"reference": FR_SEMANTICS_REFERENCE, "sparkle": FR_SEMANTICS_SPARKLE}
FR_SHARD_NONE, FR_SHARD_VOICES, FR_SHARD_PARTIALS = 0, 1, 2
SHARD_MODES = {"none": FR_SHARD_NONE, "voices": FR_SHARD_VOICES, "partials": FR_SHARD_PARTIALS}
FR_SHARD_GATHER = 1
FR_SHARD_SERIAL_EXCHANGE = 2
FR_COMM_ID_BYTES = 128
FR_CONFIG_SYNC_COMPILE = 1

FR_ABI_VERSION = 2

EDGE_DTYPE = np.dtype([("from", "<u4"), ("to", "<u4"), ("from_slot", "<u4"), ("to_slot", "<u4")])


class fr_edge(C.Structure):
    _fields_ = [("from_", C.c_uint32), ("to", C.c_uint32), ("from_slot", C.c_uint32), ("to_slot", C.c_uint32)]


class fr_effect(C.Structure):
    pass


fr_effect._fields_ = [
    ("kind", C.c_int32),
    ("n_nodes", C.c_uint32),
    ("node_handles", C.POINTER(C.c_uint32)),
    ("node_effects", C.POINTER(C.POINTER(fr_effect))),
    ("n_edges", C.c_uint32),
    ("edges", C.POINTER(fr_edge)),
]


class fr_config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("mode", C.c_int32), ("flags", C.c_uint32),
                ("semantics", C.c_int32), ("reserved", C.c_uint32), ("history_frames", C.c_uint64)]


SENDRECV_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)


class fr_comm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("sendrecv", SENDRECV_FN)]


class fr_shard(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("mode", C.c_int32), ("flags", C.c_uint32),
                ("rccl_id", C.POINTER(C.c_uint8)), ("comm", C.POINTER(fr_comm))]


class RenderError(RuntimeError):
    """A non-zero fr_status.  The reference panics where these are raised (reference.rs asserts)."""

    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__(f"fr_status {status} ({what}){': ' + detail if detail else ''}")


def f32_bits(x):
    """`f32::to_bits`: how a constant rides on an F32Constant edge's from_slot (effect.rs:390-417)."""
    return int(np.float32(x).view(np.uint32))


class Effect:
    """Owns a ctypes fr_effect tree (and keeps its arrays alive)."""

    _prim_cache = {}

    def __init__(self, kind, nodes=(), edges=()):
        self.kind = kind
        self._nodes = [(int(h), e) for h, e in nodes]
        self._children = [e for _, e in nodes]
        self.c = fr_effect()
        self.c.kind = kind
        n = len(nodes)
        self._handles = (C.c_uint32 * max(n, 1))(*[h for h, _ in nodes])
        self._effects = (C.POINTER(fr_effect) * max(n, 1))(*[C.pointer(e.c) for _, e in nodes])
        edges = np.ascontiguousarray(np.asarray(edges, dtype=np.uint32).reshape(-1, 4))
        self._edges = edges
        self.c.n_nodes = n
        self.c.node_handles = C.cast(self._handles, C.POINTER(C.c_uint32))
        self.c.node_effects = C.cast(self._effects, C.POINTER(C.POINTER(fr_effect)))
        self.c.n_edges = len(edges)
        self.c.edges = edges.ctypes.data_as(C.POINTER(fr_edge))

    @classmethod
    def primitive(cls, name):
        if name not in cls._prim_cache:
            cls._prim_cache[name] = cls(FR_PRIM[name])
        return cls._prim_cache[name]

    @classmethod
    def graph(cls, nodes, edges):
        """nodes: [(handle, Effect)], edges: [(from, to, from_slot, to_slot)] -- an AdjList (adjlist.rs:11-15)."""
        return cls(FR_EFFECT_GRAPH, list(nodes), list(edges))

    def to_json(self):
        """Inverse of from_json (fixture form)."""
        if self.kind != FR_EFFECT_GRAPH:
            return PRIMITIVES[self.kind]
        return {"graph": {"nodes": [[h, e.to_json()] for h, e in self._nodes],
                          "edges": [[int(x) for x in row] for row in self._edges]}}

    @classmethod
    def from_json(cls, spec):
        """'Delay' | {'graph': {'nodes': [[h, spec]...], 'edges': [[f,t,fs,ts]...]}} (fixture form)."""
        if isinstance(spec, str):
            return cls.primitive(spec)
        g = spec["graph"]
        return cls.graph([(h, cls.from_json(s)) for h, s in g["nodes"]], g["edges"])


class RendererLib:
    """A loaded shared library exporting the fr_* ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found. Build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback for the HIP engine.")
        self.path = path
        L = self.lib = C.CDLL(path)
        P = C.POINTER
        vp = C.c_void_p
        L.fr_renderer_create.argtypes = [P(fr_config), P(vp)]
        L.fr_renderer_create.restype = C.c_int32
        L.fr_renderer_destroy.argtypes = [vp]
        L.fr_renderer_destroy.restype = None
        L.fr_on_add_node.argtypes = [vp, C.c_uint32, P(fr_effect)]
        L.fr_on_del_node.argtypes = [vp, C.c_uint32]
        L.fr_on_add_edge.argtypes = [vp, P(fr_edge)]
        L.fr_on_del_edge.argtypes = [vp, P(fr_edge)]
        L.fr_on_add_nodes.argtypes = [vp, P(C.c_uint32), P(P(fr_effect)), C.c_size_t]
        L.fr_on_add_edges.argtypes = [vp, vp, C.c_size_t]
        L.fr_fill_buffer.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp, C.c_uint32]
        L.fr_fill_buffer_device.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp, C.c_uint32, vp]
        L.fr_set_track_inputs.argtypes = [vp, C.c_uint32]
        L.fr_fill_buffer_dense.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, vp, C.c_uint32]
        L.fr_fill_buffer_device_dense.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, vp, C.c_uint32, vp]
        for fn in ("fr_set_track_inputs", "fr_fill_buffer_dense", "fr_fill_buffer_device_dense"):
            getattr(L, fn).restype = C.c_int32
        for fn in ("fr_on_add_node", "fr_on_del_node", "fr_on_add_edge", "fr_on_del_edge", "fr_on_add_nodes",
                   "fr_on_add_edges", "fr_fill_buffer", "fr_fill_buffer_device", "fr_set_timing",
                   "fr_get_timing", "fr_reset_timing"):
            getattr(L, fn).restype = C.c_int32
        L.fr_last_error.argtypes = [vp]
        L.fr_last_error.restype = C.c_char_p
        L.fr_status_string.argtypes = [C.c_int32]
        L.fr_status_string.restype = C.c_char_p
        L.fr_backend_name.argtypes = []
        L.fr_backend_name.restype = C.c_char_p
        L.fr_abi_version.argtypes = []
        L.fr_abi_version.restype = C.c_uint32
        L.fr_plan_json.argtypes = [vp]
        L.fr_plan_json.restype = C.c_char_p
        L.fr_set_timing.argtypes = [vp, C.c_int32]
        L.fr_get_timing.argtypes = [vp, C.c_char_p, P(C.c_double), P(C.c_uint64)]
        L.fr_reset_timing.argtypes = [vp]
        L.fr_set_shard.argtypes = [vp, P(fr_shard)]
        L.fr_set_shard.restype = C.c_int32
        L.fr_shard_rows.argtypes = [vp, C.c_uint32, P(C.c_uint32), P(C.c_uint32)]
        L.fr_shard_rows.restype = C.c_int32
        L.fr_host_register.argtypes = [vp, vp, C.c_size_t]
        L.fr_host_register.restype = C.c_int32
        L.fr_host_unregister.argtypes = [vp, vp]
        L.fr_host_unregister.restype = C.c_int32
        L.fr_stream_begin.argtypes = [C.c_void_p, C.c_uint32]
        L.fr_stream_begin.restype = C.c_int32
        L.fr_stream_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
        L.fr_stream_block.restype = C.c_int32
        L.fr_stream_end.argtypes = [C.c_void_p]
        L.fr_stream_end.restype = C.c_int32
        L.fr_comm_selftest.argtypes = [C.c_int32, C.c_uint64]
        L.fr_comm_selftest.restype = C.c_int32
        L.fr_comm_unique_id.argtypes = [P(C.c_uint8)]
        L.fr_comm_unique_id.restype = C.c_int32
        if L.fr_abi_version() != FR_ABI_VERSION:
            raise RuntimeError(f"{path}: ABI version {L.fr_abi_version()} != {FR_ABI_VERSION}")

    @property
    def backend(self):
        return self.lib.fr_backend_name().decode()

    def status_string(self, s):
        return self.lib.fr_status_string(s).decode()

    def comm_selftest(self, n_floats=1 << 16, device=-1):
        """One exchange through the RCCL transport with this rank as its own peer (fr_comm_selftest)."""
        st = self.lib.fr_comm_selftest(device, n_floats)
        if st != FR_OK:
            raise RenderError(st, self.status_string(st), "fr_comm_selftest")

    def comm_unique_id(self):
        """ncclGetUniqueId through the engine: 128 bytes to hand to every rank's set_shard(rccl_id=...)."""
        buf = (C.c_uint8 * FR_COMM_ID_BYTES)()
        st = self.lib.fr_comm_unique_id(buf)
        if st != FR_OK:
            raise RenderError(st, self.status_string(st), "fr_comm_unique_id")
        return bytes(buf)


class Renderer:
    """Handle-owning wrapper: one method per entry point, arguments as in the reference's traits."""

    def __init__(self, rlib, mode="auto", device=-1, semantics="reference", history_frames=0, sync_compile=True):
        """sync_compile: hipRTC specialisations are compiled inside the call that first needs them (deterministic plans:
        what tests and benchmarks want); False = the ABI's default, compile on a worker thread and switch over when ready."""
        self.rlib = rlib
        self.L = rlib.lib
        self._stream_slots = 0
        cfg = fr_config(FR_ABI_VERSION, device, MODES[mode] if isinstance(mode, str) else mode, FR_CONFIG_SYNC_COMPILE if sync_compile else 0,
                        SEMANTICS[semantics] if isinstance(semantics, str) else semantics, 0, history_frames)
        h = C.c_void_p()
        st = self.L.fr_renderer_create(C.byref(cfg), C.byref(h))
        if st != FR_OK:
            raise RenderError(st, rlib.status_string(st), "fr_renderer_create")
        self.h = h
        self._offs1 = np.zeros(2, dtype=np.uint64)
        self._keep = []  # effects passed in stay alive as long as the renderer (not required by the ABI)

    def close(self):
        if getattr(self, "h", None):
            self.L.fr_renderer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st):
        if st != FR_OK:
            raise RenderError(st, self.rlib.status_string(st), self.L.fr_last_error(self.h).decode())

    # --- block streaming (fr_stream_*): one resident launch serves blocks of up to 64 frames ---
    def stream_begin(self, n_slots):
        self._check(self.L.fr_stream_begin(self.h, n_slots))
        self._stream_slots = n_slots

    def stream_block(self, start, row, out=None):
        """Render the frames [start, start + len(row)) of every slot from the block's input row (float32)."""
        row = np.ascontiguousarray(row, dtype=np.float32)
        n = len(row)
        if out is None:
            out = np.empty((self._stream_slots, n), dtype=np.float32)
        # (the C call takes no slot count: it writes stream_slots rows of n floats, so the buffer is checked here)
        if out.dtype != np.float32 or not out.flags["C_CONTIGUOUS"] or out.shape != (self._stream_slots, n):
            raise ValueError(f"stream_block: out must be a C-contiguous float32 array of shape ({self._stream_slots}, {n}), got {out.dtype} {out.shape}")
        self._check(self.L.fr_stream_block(self.h, out.ctypes.data, n, start, row.ctypes.data, n))
        return out

    def stream_end(self):
        self._check(self.L.fr_stream_end(self.h))

    def host_register(self, array):
        """Page-lock a numpy buffer the caller reuses as `out=` / input rows (fr_host_register)."""
        self._check(self.L.fr_host_register(self.h, array.ctypes.data, array.nbytes))

    def host_unregister(self, array):
        self._check(self.L.fr_host_unregister(self.h, array.ctypes.data))

    # --- sharding (fr_set_shard) ---
    def set_shard(self, rank, world, mode="voices", gather=False, rccl_id=None, sendrecv=None, serial_exchange=False):
        """This renderer becomes rank `rank` of `world`.  Transport of the exchange step: `rccl_id` (bytes from
        RendererLib.comm_unique_id(), the same on every rank: the engine's own RCCL communicator) or `sendrecv`, a
        Python callable (peer, send: np.uint8 array view or None, recv: np.uint8 array view or None) -> None that
        exchanges HOST buffers with `peer` (the engine stages device ranges through pinned memory around it)."""
        sh = fr_shard()
        sh.rank, sh.world = rank, world
        sh.mode = SHARD_MODES[mode] if isinstance(mode, str) else mode
        sh.flags = (FR_SHARD_GATHER if gather else 0) | (FR_SHARD_SERIAL_EXCHANGE if serial_exchange else 0)
        keep = []
        if rccl_id is not None:
            idbuf = (C.c_uint8 * FR_COMM_ID_BYTES).from_buffer_copy(rccl_id)
            sh.rccl_id = C.cast(idbuf, C.POINTER(C.c_uint8))
            keep.append(idbuf)
        if sendrecv is not None:
            def thunk(_ctx, peer, send, send_bytes, recv, recv_bytes):
                try:
                    sv = np.ctypeslib.as_array((C.c_uint8 * send_bytes).from_address(send)) if send_bytes else None
                    rv = np.ctypeslib.as_array((C.c_uint8 * recv_bytes).from_address(recv)) if recv_bytes else None
                    sendrecv(int(peer), sv, rv)
                    return 0
                except Exception as e:  # noqa: BLE001 -- an exception must not unwind through the C frames
                    self._comm_error = e
                    return 1
            cb = SENDRECV_FN(thunk)
            comm = fr_comm(None, cb)
            sh.comm = C.pointer(comm)
            keep += [cb, comm]
        self._shard_keep = keep   # the callback must outlive every later fill_buffer
        self._check(self.L.fr_set_shard(self.h, C.byref(sh)))

    def shard_rows(self, n_slots):
        lo, hi = C.c_uint32(), C.c_uint32()
        self._check(self.L.fr_shard_rows(self.h, n_slots, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    # --- GraphWatcher (graphwatcher.rs:4-9) ---
    def on_add_node(self, handle, effect):
        if isinstance(effect, (str, dict)):
            effect = Effect.from_json(effect)
        self._check(self.L.fr_on_add_node(self.h, handle, C.byref(effect.c)))

    def on_del_node(self, handle):
        self._check(self.L.fr_on_del_node(self.h, handle))

    def on_add_edge(self, frm, to, from_slot, to_slot):
        e = fr_edge(frm, to, from_slot, to_slot)
        self._check(self.L.fr_on_add_edge(self.h, C.byref(e)))

    def on_del_edge(self, frm, to, from_slot, to_slot):
        e = fr_edge(frm, to, from_slot, to_slot)
        self._check(self.L.fr_on_del_edge(self.h, C.byref(e)))

    # --- batch forms ---
    def on_add_nodes(self, handles, effects):
        """handles: uint32 array; effects: one Effect for all, or a sequence of Effects."""
        handles = np.ascontiguousarray(handles, dtype=np.uint32)
        n = len(handles)
        if isinstance(effects, Effect):
            arr = np.full(n, C.addressof(effects.c), dtype=np.uint64)
            ptrs = arr.ctypes.data_as(C.POINTER(C.POINTER(fr_effect)))
            self._keep.append(effects)
        else:
            effects = list(effects)
            arr = np.fromiter((C.addressof(e.c) for e in effects), dtype=np.uint64, count=n)
            ptrs = arr.ctypes.data_as(C.POINTER(C.POINTER(fr_effect)))
            self._keep.extend(effects)
        self._check(self.L.fr_on_add_nodes(self.h, handles.ctypes.data_as(C.POINTER(C.c_uint32)), ptrs, n))

    def on_add_edges(self, edges):
        """edges: uint32 array [n,4] of (from, to, from_slot, to_slot)."""
        edges = np.ascontiguousarray(edges, dtype=np.uint32).reshape(-1, 4)
        self._check(self.L.fr_on_add_edges(self.h, edges.ctypes.data, len(edges)))

    # --- Renderer::fill_buffer (renderer.rs:16) ---
    def fill_buffer(self, n_slots, start, end, inputs=(), out=None):
        """Render [start, end) for output slots 0..n_slots; inputs = rows of a Jagged2<f32>."""
        n_times = end - start
        if out is None:
            out = np.zeros((n_slots, n_times), dtype=np.float32)  # Dispatch allocates zeros (dispatch.rs:149)
        assert out.dtype == np.float32 and out.shape == (n_slots, n_times) and out.flags.c_contiguous
        if (len(inputs) == 1 and isinstance(inputs[0], np.ndarray) and inputs[0].dtype == np.float32 and inputs[0].ndim == 1
                and inputs[0].flags.c_contiguous and len(inputs[0])):
            # the usual call (one ready-made row): no marshalling copies, so that timing this method times the library
            data = inputs[0]
            offs = self._offs1
            offs[1] = len(data)
            n_rows = 1
        else:
            rows = [np.ascontiguousarray(r, dtype=np.float32).ravel() for r in inputs]
            offs = np.zeros(len(rows) + 1, dtype=np.uint64)
            if rows:
                offs[1:] = np.cumsum([len(r) for r in rows])
                data = np.concatenate(rows) if offs[-1] else np.zeros(1, np.float32)
            else:
                data = np.zeros(1, np.float32)
            n_rows = len(rows)
        self._check(self.L.fr_fill_buffer(self.h, out.ctypes.data, n_slots, n_times, start,
                                          data.ctypes.data, offs.ctypes.data, n_rows))
        return out

    # --- control-rate tracks (friendship_render.h): input slots >= first_slot are read in place, never stored ---
    def set_track_inputs(self, first_slot):
        self._check(self.L.fr_set_track_inputs(self.h, 0xFFFFFFFF if first_slot is None else first_slot))

    def fill_buffer_dense(self, n_slots, start, end, inputs, out=None):
        """fill_buffer with the reference's own input shape: inputs = float32 [n_in_rows, end - start] (row r feeds slot r)."""
        n_times = end - start
        inputs = np.ascontiguousarray(inputs, dtype=np.float32)
        assert inputs.ndim == 2 and (inputs.shape[1] == n_times or inputs.shape[0] == 0)
        if out is None:
            out = np.zeros((n_slots, n_times), dtype=np.float32)
        assert out.dtype == np.float32 and out.shape == (n_slots, n_times) and out.flags.c_contiguous
        self._check(self.L.fr_fill_buffer_dense(self.h, out.ctypes.data, n_slots, n_times, start, inputs.ctypes.data, inputs.shape[0]))
        return out

    def fill_buffer_device_dense(self, d_out_ptr, n_slots, n_times, idx, d_in_ptr, n_in_rows, stream=0):
        self._check(self.L.fr_fill_buffer_device_dense(self.h, d_out_ptr, n_slots, n_times, idx, d_in_ptr, n_in_rows, stream))

    def fill_buffer_device(self, d_out_ptr, n_slots, n_times, idx, d_in_ptr, row_offsets, stream=0):
        """Device-resident variant: raw device pointers (e.g. torch.Tensor.data_ptr()), host offsets."""
        offs = np.ascontiguousarray(row_offsets, dtype=np.uint64)
        n_rows = len(offs) - 1 if len(offs) else 0
        self._check(self.L.fr_fill_buffer_device(self.h, d_out_ptr, n_slots, n_times, idx, d_in_ptr,
                                                 offs.ctypes.data if n_rows else None, n_rows, stream))

    # --- introspection ---
    def plan(self):
        s = self.L.fr_plan_json(self.h)
        return json.loads(s.decode()) if s else {}

    def set_timing(self, on=True):
        self._check(self.L.fr_set_timing(self.h, 1 if on else 0))

    def reset_timing(self):
        self._check(self.L.fr_reset_timing(self.h))

    def get_timing(self, kernel_class="all"):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.L.fr_get_timing(self.h, kernel_class.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
