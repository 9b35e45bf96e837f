"""
ORACLE (test infrastructure) -- ctypes binding of oracle/libzkattest_oracle.so, the C restatement.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libzkattest_oracle.so')


def build(force=False):
    src = os.path.join(_HERE, 'zkattest_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.zko_ctx_create.restype = C.c_void_p
        L.zko_ctx_destroy.argtypes = [C.c_void_p]
        L.zko_ctx_set_params.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32]
        L.zko_ctx_set_ring.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.zko_max_proof_size.argtypes = [C.c_void_p]
        L.zko_max_proof_size.restype = C.c_uint64
        L.zko_prove_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int,
                                      C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        L.zko_verify_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
        L.zko_p256_mul.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.zko_tom_mul.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.zko_tom_commit.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.zko_field_op.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
        L.zko_sha256.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p]
        _lib = L
    return _lib


class OracleCtx:
    def __init__(self, nist_h64, tom_g72, tom_h72, sec_level=80):
        self.L = lib()
        self.h = self.L.zko_ctx_create()
        rc = self.L.zko_ctx_set_params(self.h, bytes(nist_h64), bytes(tom_g72), bytes(tom_h72), sec_level)
        if rc:
            raise ValueError('zko_ctx_set_params status %d' % rc)

    def set_ring(self, keys_be32, nkeys):
        rc = self.L.zko_ctx_set_ring(self.h, bytes(keys_be32), nkeys)
        if rc:
            raise ValueError('zko_ctx_set_ring status %d' % rc)

    def prove_batch(self, msg, sig, pk, which, seeds=None, streams=None, stream_blocks=0, nthreads=1):
        """Returns (list of proof bytes (None on error), list of status)."""
        B = len(which)
        slot = self.L.zko_max_proof_size(self.h)
        out = C.create_string_buffer(slot * B)
        sizes = (C.c_uint64 * B)()
        status = (C.c_int32 * B)()
        w = (C.c_uint32 * B)(*which)
        if streams is None:
            mode, data, stride = 0, bytes(seeds), 0
        else:
            mode, data, stride = 1, bytes(streams), stream_blocks
        rc = self.L.zko_prove_batch(self.h, B, bytes(msg), bytes(sig), bytes(pk), w, mode, data, stride, out, slot, sizes, status, nthreads)
        if rc:
            raise ValueError('zko_prove_batch status %d' % rc)
        raw = out.raw
        proofs = [raw[slot * b: slot * b + sizes[b]] if status[b] == 0 else None for b in range(B)]
        return proofs, list(status)

    def verify_batch(self, msg, proofs, nthreads=1, vseeds=None):
        B = len(proofs)
        off = (C.c_uint64 * (B + 1))()
        o = 0
        for b, p in enumerate(proofs):
            off[b] = o
            o += len(p)
        off[B] = o
        ok = (C.c_uint8 * B)()
        status = (C.c_int32 * B)()
        rc = self.L.zko_verify_batch(self.h, B, bytes(msg), b''.join(proofs), off, bytes(vseeds) if vseeds is not None else None, ok, status, nthreads)
        if rc:
            raise ValueError('zko_verify_batch status %d' % rc)
        return list(ok), list(status)

    def tom_commit(self, v, r):
        out = C.create_string_buffer(72)
        self.L.zko_tom_commit(self.h, v.to_bytes(32, 'big'), r.to_bytes(32, 'big'), out)
        return out.raw

    def close(self):
        if self.h:
            self.L.zko_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def p256_mul(k, p_xy=None):
    out = C.create_string_buffer(64)
    rc = lib().zko_p256_mul(k.to_bytes(32, 'big'), p_xy, out)
    return out.raw if rc == 1 else None


def tom_mul(k, p_xy=None):
    out = C.create_string_buffer(72)
    rc = lib().zko_tom_mul(k.to_bytes(32, 'big'), p_xy, out)
    assert rc == 1
    return out.raw


def field_op(which, op, a, b=0):
    out = C.create_string_buffer(40)
    rc = lib().zko_field_op(which, op, a.to_bytes(40, 'big'), b.to_bytes(40, 'big'), out)
    assert rc == 0
    return int.from_bytes(out.raw, 'big')


def sha256(data):
    out = C.create_string_buffer(32)
    lib().zko_sha256(bytes(data), len(data), out)
    return out.raw
