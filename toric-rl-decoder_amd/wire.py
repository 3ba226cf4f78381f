"""Wire format of packed transition blocks, host side (layout: include/toricenv.h).

The device writes the blocks (tq_actor_step); this module is what the replay-memory process on
the host uses to ingest a gathered block without per-item Python: it views the
sections of the byte buffer and expands them into the reference's record type
(``transition_type``, Actor_mp.py:52-56) in bulk with numpy.
"""
import numpy as np


def words(d):
    return (d * d + 63) // 64


def _align8(x):
    return (x + 7) & ~7


def block_bytes(d, cap):
    w = words(d)
    return 4 * 8 * w * cap + 3 * _align8(4 * cap) + _align8(cap)


def sections(d, cap):
    """name -> (byte offset, numpy dtype, shape) of each SoA section."""
    w = words(d)
    out, off = {}, 0
    for name in ("persp_v", "persp_p", "next_v", "next_p"):
        out[name] = (off, np.uint64, (w, cap))
        off += 8 * w * cap
    out["action"] = (off, np.uint32, (cap,))
    off += _align8(4 * cap)
    out["reward"] = (off, np.float32, (cap,))
    off += _align8(4 * cap)
    out["priority"] = (off, np.float32, (cap,))
    off += _align8(4 * cap)
    out["terminal"] = (off, np.uint8, (cap,))
    return out


def view(buf, d, cap):
    """Zero-copy numpy views of the sections of a block held in a uint8 buffer."""
    buf = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    if buf.size < block_bytes(d, cap):
        raise ValueError("buffer smaller than the block")
    out = {}
    for name, (off, dt, shape) in sections(d, cap).items():
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        out[name] = buf[off:off + n].view(dt).reshape(shape)
    return out


def _unpack_planes(v, p, d):
    """u64[W][n] vertex/plaquette planes -> u8[n,2,d,d] (bit r*d+c of a plane = cell (r,c))."""
    n = v.shape[1]
    dd = d * d
    out = np.empty((n, 2, dd), np.uint8)
    for k, plane in enumerate((v, p)):
        bits = np.unpackbits(np.ascontiguousarray(plane.T).view(np.uint8), axis=1, bitorder="little")
        out[:, k, :] = bits[:, :dd]
    return out.reshape(n, 2, d, d)


def decode(buf, d, cap, first=0, count=None, drop_empty=True):
    """Slots [first, first+count) -> dict(perspective u8[n,2,d,d], next_perspective, action i32[n,4],
    reward f32[n], terminal bool[n], priority f32[n], slot i64[n]).  A slot whose action word is 0
    (op = 0) holds no transition (include/toricenv.h); such slots are dropped unless
    ``drop_empty=False``; ``slot`` gives the block slot of every returned row."""
    count = cap - first if count is None else count
    s = view(buf, d, cap)
    sl = np.arange(first, first + count)
    if drop_empty:
        sl = sl[(s["action"][first:first + count] >> 24) != 0]
    a = s["action"][sl]
    action = np.stack((a & 255, (a >> 8) & 255, (a >> 16) & 255, a >> 24), axis=1).astype(np.int32)
    return dict(perspective=_unpack_planes(s["persp_v"][:, sl], s["persp_p"][:, sl], d),
                next_perspective=_unpack_planes(s["next_v"][:, sl], s["next_p"][:, sl], d),
                action=action, reward=s["reward"][sl].copy(), terminal=s["terminal"][sl].astype(bool),
                priority=s["priority"][sl].copy(), slot=sl)


def encode(d, perspective, next_perspective, action, reward, terminal, cap=None, priority=None):
    """Inverse of decode (used by tests and by hosts that need to re-pack records)."""
    n = perspective.shape[0]
    cap = n if cap is None else cap
    buf = np.zeros(block_bytes(d, cap), np.uint8)
    s = view(buf, d, cap)
    dd, w = d * d, words(d)
    for (nv, np_), grid in ((("persp_v", "persp_p"), perspective), (("next_v", "next_p"), next_perspective)):
        g = np.asarray(grid, np.uint8).reshape(n, 2, dd)
        for k, name in enumerate((nv, np_)):
            bits = np.zeros((n, 64 * w), np.uint8)
            bits[:, :dd] = g[:, k, :] != 0
            packed = np.packbits(bits, axis=1, bitorder="little").view(np.uint64)     # (n, w)
            s[name][:, :n] = packed.T
    a = np.asarray(action, np.uint32).reshape(n, 4)
    s["action"][:n] = a[:, 0] | (a[:, 1] << 8) | (a[:, 2] << 16) | (a[:, 3] << 24)
    s["reward"][:n] = np.asarray(reward, np.float32)
    s["terminal"][:n] = np.asarray(terminal).astype(np.uint8)
    if priority is not None:
        s["priority"][:n] = np.asarray(priority, np.float32)
    return buf


def transition_type(d):
    """The reference's replay record (Actor_mp.py:52-56, util.py:10)."""
    action_type = np.dtype([('position', (np.int64, 3)), ('op', np.int64)])
    return np.dtype([('perspective', (np.int64, (2, d, d))), ('action', action_type), ('reward', np.float64),
                     ('next_perspective', (np.int64, (2, d, d))), ('terminal', np.bool_)])


def to_records(decoded, d):
    """dict from decode() -> (records, priorities): a numpy array of the reference's transition_type
    (Actor_mp.py:52-56) and the f32 priority of each record -- the (transition, priority) pairs the
    replay process saves one by one (IO_mp.py:60-66); ``zip(records, priorities)`` is the list the
    actor puts on its queue (Actor_mp.py:152)."""
    n = decoded["perspective"].shape[0]
    rec = np.empty(n, transition_type(d))
    rec['perspective'] = decoded["perspective"]
    rec['next_perspective'] = decoded["next_perspective"]
    rec['action']['position'] = decoded["action"][:, :3]
    rec['action']['op'] = decoded["action"][:, 3]
    rec['reward'] = decoded["reward"]
    rec['terminal'] = decoded["terminal"]
    return rec, np.asarray(decoded["priority"], np.float32)
