"""ctypes binding of libzkattest_hip.so (C ABI: include/zkattest.h).  Fails loudly when the library is missing:
there is no CPU fallback in the product path."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZKATTEST_LIB selects another build of the same library (e.g. a different comb width, csrc/Makefile TOM_WIN_BITS)
LIB_PATH = os.environ.get('ZKATTEST_LIB') or os.path.join(_HERE, 'lib', 'libzkattest_hip.so')
CSRC = os.path.join(_HERE, 'csrc')

# every symbol include/zkattest.h declares
SYMBOLS = [
    'zk_ctx_create', 'zk_ctx_destroy', 'zk_strerror', 'zk_last_error', 'zk_ctx_set_params', 'zk_ctx_set_ring',
    'zk_ctx_set_ring_device', 'zk_ctx_wipe', 'zk_last_wall_ms', 'zk_ctx_set_timing', 'zk_keys_to_ints', 'zk_ctx_set_chunk', 'zk_ctx_set_lanes', 'zk_ctx_set_comb_bits', 'zk_ctx_set_batch_verify', 'zk_proof_max_size', 'zk_prove_batch', 'zk_prove_batch_device',
    'zk_verify_batch', 'zk_verify_batch_device', 'zk_synth_workload', 'zk_synth_params', 'zk_last_timing',
    'zk_proof_to_json', 'zk_proof_from_json', 'zk_host_alloc', 'zk_host_free', 'zk_ctx_set_host_taper', 'zk_ctx_set_slice', 'zk_ctx_set_mode', 'zk_ring_digest', 'zk_hardened_h',
    'zk_pool_create', 'zk_pool_destroy', 'zk_pool_size', 'zk_pool_ctx', 'zk_pool_last_error', 'zk_pool_ring_transport', 'zk_pool_rccl_library', 'zk_pool_shard',
    'zk_pool_set_params', 'zk_pool_set_ring', 'zk_pool_prove_batch', 'zk_pool_verify_batch',
    'zk_pool_host_alloc', 'zk_pool_host_free', 'zk_pool_numa_node', 'zk_pool_test_locality', 'zk_pool_shard_ms', 'zk_ctx_copy_probe', 'zk_ctx_set_wire', 'zk_proof_pack', 'zk_proof_unpack', 'zk_pool_prove_batch_device', 'zk_pool_device_alloc', 'zk_pool_device_free',
    'zk_prove_submit', 'zk_prove_submit_device', 'zk_prove_wait', 'zk_verify_submit', 'zk_verify_wait', 'zk_test_counter', 'zk_ctx_set_key_tables',
    'zk_proofs_to_json_batch', 'zk_proofs_from_json_batch', 'zk_ctx_set_ring_fold',
    'zk_pool_prove_submit', 'zk_pool_prove_wait', 'zk_pool_verify_submit', 'zk_pool_verify_wait', 'zk_ctx_set_verify_groups',
    'zk_test_field_op', 'zk_test_tom_commit', 'zk_test_p256_fixed_mul', 'zk_test_sha256', 'zk_test_rng_draws',
]

STATUS_TEXT = {
    0: 'ok', 1: 'point not in group', 2: 'invalid public key', 3: 'T[i] is at infinity', 4: 'T1 is at infinity',
    5: 'P/Q/R is at infinity', 6: "Points don't add up!", 7: 'R is at infinity', 8: 'params not found',
    9: 'security level not achieved', 10: 'error deserializing', 11: 'randomness stream exhausted',
    12: 'buffer too small or context not configured', 13: 'incorrect interpolation', 14: 'invalid argument',
    15: 'HIP runtime failure',
}


class ZkRng(C.Structure):
    _fields_ = [('mode', C.c_int32), ('data', C.c_void_p), ('stride_blocks', C.c_uint64)]


def build(jobs=8):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(['make', '-C', CSRC, '-j%d' % jobs, '-s'])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libzkattest_hip.so is not built (run __graft_entry__.build()); the engine has no CPU fallback')
        L = C.CDLL(LIB_PATH)
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        L.zk_ctx_create.argtypes = [i32, C.POINTER(vp)]
        L.zk_ctx_destroy.argtypes = [vp]
        L.zk_ctx_destroy.restype = None
        L.zk_strerror.argtypes = [i32]
        L.zk_strerror.restype = C.c_char_p
        L.zk_last_error.argtypes = [vp]
        L.zk_last_error.restype = C.c_char_p
        L.zk_ctx_set_params.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, u32]
        L.zk_ctx_set_ring.argtypes = [vp, C.c_char_p, u64]
        L.zk_ctx_set_ring_device.argtypes = [vp, vp, u64]
        L.zk_keys_to_ints.argtypes = [vp, u64, C.c_char_p, vp, vp]
        L.zk_ctx_set_chunk.argtypes = [vp, u32]
        L.zk_ctx_set_lanes.argtypes = [vp, u32]
        L.zk_ctx_set_comb_bits.argtypes = [vp, u32]
        L.zk_ctx_set_batch_verify.argtypes = [vp, u32]
        L.zk_proof_max_size.argtypes = [vp]
        L.zk_proof_max_size.restype = u64
        L.zk_prove_batch.argtypes = [vp, u64, C.c_char_p, C.c_char_p, C.c_char_p, vp, C.POINTER(ZkRng), vp, u64, vp, vp]
        L.zk_prove_batch_device.argtypes = [vp, u64, vp, vp, vp, vp, C.POINTER(ZkRng), vp, u64, vp, vp]
        L.zk_verify_batch.argtypes = [vp, u64, C.c_char_p, vp, vp, C.c_char_p, vp, vp]
        L.zk_host_alloc.argtypes = [C.c_size_t]
        L.zk_host_alloc.restype = vp
        L.zk_host_free.argtypes = [vp]
        L.zk_host_free.restype = None
        L.zk_verify_batch_device.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp]
        L.zk_synth_workload.argtypes = [vp, u64, u64, u64, vp, vp, vp, vp, vp, vp]
        L.zk_synth_params.argtypes = [vp, u64, vp, vp, vp]
        L.zk_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_char_p), C.POINTER(C.c_float), u32]
        L.zk_last_timing.restype = u32
        L.zk_proof_to_json.argtypes = [C.c_char_p, u64, vp, u64, C.POINTER(u64)]
        L.zk_proof_from_json.argtypes = [C.c_char_p, u64, vp, u64, C.POINTER(u64)]
        L.zk_ctx_set_host_taper.argtypes = [vp, u32]
        L.zk_ctx_set_slice.argtypes = [vp, u32]
        L.zk_ctx_set_mode.argtypes = [vp, u32]
        L.zk_ring_digest.argtypes = [vp, vp]
        L.zk_hardened_h.argtypes = [C.c_char_p, u64, vp, vp]
        L.zk_pool_create.argtypes = [C.POINTER(C.c_int), i32, C.POINTER(vp)]
        L.zk_pool_destroy.argtypes = [vp]
        L.zk_pool_destroy.restype = None
        L.zk_pool_size.argtypes = [vp]
        L.zk_pool_ctx.argtypes = [vp, i32]
        L.zk_pool_ctx.restype = vp
        L.zk_pool_last_error.argtypes = [vp]
        L.zk_pool_last_error.restype = C.c_char_p
        L.zk_pool_ring_transport.argtypes = [vp]
        L.zk_pool_ring_transport.restype = C.c_char_p
        L.zk_pool_rccl_library.argtypes = [vp]
        L.zk_pool_rccl_library.restype = C.c_char_p
        L.zk_pool_shard.argtypes = [vp, u64, i32, C.POINTER(u64), C.POINTER(u64)]
        L.zk_pool_shard.restype = None
        L.zk_pool_set_params.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, u32]
        L.zk_pool_set_ring.argtypes = [vp, C.c_char_p, u64]
        L.zk_pool_prove_batch.argtypes = [vp, u64, C.c_char_p, C.c_char_p, C.c_char_p, vp, C.POINTER(ZkRng), vp, u64, vp, vp, vp]
        L.zk_pool_verify_batch.argtypes = [vp, u64, C.c_char_p, vp, vp, vp, C.c_char_p, vp, vp]
        L.zk_proofs_to_json_batch.argtypes = [u64, vp, vp, vp, u64, vp, vp, u32]
        L.zk_proofs_from_json_batch.argtypes = [u64, vp, vp, vp, u64, vp, vp, u32]
        L.zk_prove_submit.argtypes = [vp, u64, C.c_char_p, C.c_char_p, C.c_char_p, vp, C.POINTER(ZkRng), vp, u64, vp, vp, C.POINTER(vp)]
        L.zk_prove_submit_device.argtypes = [vp, u64, vp, vp, vp, vp, C.POINTER(ZkRng), vp, u64, vp, vp, C.POINTER(vp)]
        L.zk_prove_wait.argtypes = [vp, vp]
        L.zk_verify_submit.argtypes = [vp, u64, C.c_char_p, vp, vp, C.c_char_p, vp, vp, C.POINTER(vp)]
        L.zk_verify_wait.argtypes = [vp, vp]
        L.zk_ctx_set_ring_fold.argtypes = [vp, u32]
        L.zk_ctx_set_key_tables.argtypes = [vp, u32]
        L.zk_ctx_set_verify_groups.argtypes = [vp, u32]
        L.zk_test_counter.argtypes = [vp, i32]
        L.zk_test_counter.restype = u64
        L.zk_pool_prove_submit.argtypes = [vp, u64, C.c_char_p, C.c_char_p, C.c_char_p, vp, C.POINTER(ZkRng), vp, u64, vp, vp, vp, C.POINTER(vp)]
        L.zk_pool_prove_wait.argtypes = [vp, vp]
        L.zk_pool_verify_submit.argtypes = [vp, u64, C.c_char_p, vp, vp, vp, C.c_char_p, vp, vp, C.POINTER(vp)]
        L.zk_pool_verify_wait.argtypes = [vp, vp]
        L.zk_pool_prove_batch_device.argtypes = [vp, u64, C.c_char_p, C.c_char_p, C.c_char_p, vp, C.POINTER(ZkRng), vp, vp, vp, vp, vp]
        L.zk_pool_device_alloc.argtypes = [vp, i32, C.c_size_t]
        L.zk_pool_device_alloc.restype = vp
        L.zk_pool_device_free.argtypes = [vp, i32, vp]
        L.zk_pool_device_free.restype = None
        L.zk_ctx_set_wire.argtypes = [vp, u32]
        L.zk_proof_pack.argtypes = [C.c_char_p, u64, vp, u64, C.POINTER(u64)]
        L.zk_proof_unpack.argtypes = [C.c_char_p, u64, vp, u64, C.POINTER(u64)]
        L.zk_ctx_copy_probe.argtypes = [vp, u32, C.c_size_t, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.zk_pool_host_alloc.argtypes = [vp, C.c_size_t]
        L.zk_pool_host_alloc.restype = vp
        L.zk_pool_host_free.argtypes = [vp]
        L.zk_pool_host_free.restype = None
        L.zk_pool_numa_node.argtypes = [vp, i32]
        L.zk_pool_shard_ms.argtypes = [vp, C.POINTER(C.c_float), i32]
        L.zk_pool_test_locality.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), i32]
        L.zk_test_field_op.argtypes = [vp, i32, i32, u64, C.c_char_p, C.c_char_p, vp]
        L.zk_test_tom_commit.argtypes = [vp, u64, C.c_char_p, C.c_char_p, vp]
        L.zk_test_p256_fixed_mul.argtypes = [vp, i32, u64, C.c_char_p, vp]
        L.zk_test_sha256.argtypes = [vp, u64, u64, C.c_char_p, vp]
        L.zk_test_rng_draws.argtypes = [vp, u64, C.POINTER(ZkRng), u32, u32, vp]
        _lib = L
    return _lib


class PinnedBuffer:
    """Page-locked host memory from zk_host_alloc (include/zkattest.h): .ptr for the C ABI, .view as a ctypes byte array.
    With `pool`: zk_pool_host_alloc -- the per-shard regions are placed on the NUMA nodes of the shards' devices."""

    def __init__(self, nbytes, pool=None):
        self.nbytes = nbytes
        self._pooled = pool is not None
        self.ptr = lib().zk_pool_host_alloc(pool.h, nbytes) if self._pooled else lib().zk_host_alloc(nbytes)
        if not self.ptr:
            raise MemoryError('zk_%shost_alloc(%d) failed' % ('pool_' if self._pooled else '', nbytes))
        self.view = (C.c_uint8 * nbytes).from_address(self.ptr)

    def free(self):
        if self.ptr:
            self.view = None
            (lib().zk_pool_host_free if self._pooled else lib().zk_host_free)(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ZkError(RuntimeError):
    def __init__(self, status, detail=''):
        self.status = status
        super().__init__('%s (status %d)%s' % (STATUS_TEXT.get(status, '?'), status, (': ' + detail) if detail else ''))


MODE_REFERENCE, MODE_HARDENED = 0, 1


def hardened_h(tag=b''):
    """Nothing-up-my-sleeve NistGroup.h (64 bytes) and ProofGroup.h (72 bytes) of the hardened mode (zk_hardened_h; host-only)."""
    a, b = C.create_string_buffer(64), C.create_string_buffer(72)
    rc = lib().zk_hardened_h(bytes(tag), len(tag), a, b)
    if rc:
        raise ZkError(rc)
    return a.raw, b.raw


def _wire_convert(fn, raw):
    raw = bytes(raw)
    n = C.c_uint64()
    rc = fn(raw, len(raw), None, 0, C.byref(n))
    if rc not in (0, 12):
        raise ZkError(rc, 'not a structurally complete proof of that layout')
    out = C.create_string_buffer(n.value)
    rc = fn(raw, len(raw), out, n.value, C.byref(n))
    if rc:
        raise ZkError(rc, 'not a structurally complete proof of that layout')
    return out.raw[:n.value]


def pack_proof(zka1):
    """ZKA1 -> ZKA1P (33-byte Tom coordinates): zk_proof_pack"""
    return _wire_convert(lib().zk_proof_pack, zka1)


def unpack_proof(zka1p):
    """ZKA1P -> ZKA1: zk_proof_unpack"""
    return _wire_convert(lib().zk_proof_unpack, zka1p)


def write_json(proof_bytes):
    """ZKA1 proof -> JSON text; the writeJson(SignatureProofList, proof) of src/serde.ts:34-36."""
    L = lib()
    raw = bytes(proof_bytes)
    n = C.c_uint64()
    cap = 5 * len(raw) + 4096          # the text is ~3.5x the binary proof: one call in the common case
    for _ in range(2):
        buf = C.create_string_buffer(cap)
        rc = L.zk_proof_to_json(raw, len(raw), buf, cap, C.byref(n))
        if rc != 12:
            break
        cap = n.value
    if rc:
        raise ZkError(rc)
    return buf.raw[:n.value].decode()


def read_json(text):
    """JSON text -> ZKA1 proof; the readJson(SignatureProofList, text) of src/serde.ts:21-32 (throws on bad input)."""
    L = lib()
    raw = text.encode() if isinstance(text, str) else bytes(text)
    n = C.c_uint64()
    cap = len(raw) // 2 + 64            # two hex digits per byte at least: the binary proof is shorter than half the text
    buf = C.create_string_buffer(cap)
    rc = L.zk_proof_from_json(raw, len(raw), buf, cap, C.byref(n))
    if rc == 12:
        buf = C.create_string_buffer(n.value)
        rc = L.zk_proof_from_json(raw, len(raw), buf, n.value, C.byref(n))
    if rc:
        raise ZkError(rc)
    return buf.raw[:n.value]


def _json_batch(fn, blob, off, n, cap, threads):
    L = lib()
    out_off, st = (C.c_uint64 * (n + 1))(), (C.c_int32 * n)()
    src = (C.c_uint8 * max(len(blob), 1)).from_buffer_copy(blob or b'\0')
    for _ in range(2):
        out = (C.c_uint8 * max(cap, 1))()
        rc = fn(n, src, off, out, cap, out_off, st, threads)
        if rc != 12:
            break
        cap = out_off[n]
    if rc:
        raise ZkError(rc)
    return out, out_off, st


def write_json_batch(proofs, threads=0):
    """n ZKA1 proofs -> n JSON texts on `threads` host threads (0 = all): zk_proofs_to_json_batch.  -> (texts as bytes, statuses)."""
    n = len(proofs)
    off = (C.c_uint64 * (n + 1))()
    for i, p in enumerate(proofs):
        off[i + 1] = off[i] + len(p)
    out, toff, st = _json_batch(lib().zk_proofs_to_json_batch, b''.join(proofs), off, n, int(3.7 * off[n]) + 4096 * n, threads)
    raw = memoryview(out)
    return [bytes(raw[toff[i]:toff[i + 1]]) for i in range(n)], list(st)


def read_json_batch(texts, threads=0):
    """n JSON texts -> n ZKA1 proofs (b'' where a text does not parse: see the statuses): zk_proofs_from_json_batch."""
    texts = [t.encode() if isinstance(t, str) else bytes(t) for t in texts]
    n = len(texts)
    off = (C.c_uint64 * (n + 1))()
    for i, t in enumerate(texts):
        off[i + 1] = off[i] + len(t)
    out, poff, st = _json_batch(lib().zk_proofs_from_json_batch, b''.join(texts), off, n, off[n] // 3 + 64 * n, threads)
    raw = memoryview(out)
    return [bytes(raw[poff[i]:poff[i + 1]]) for i in range(n)], list(st)


class Engine:
    """One engine = one GPU (zk_ctx)."""

    def __init__(self, device=0, _borrowed=None):
        self.L = lib()
        self.sec = None
        self._keep = []
        self._owned = _borrowed is None
        if _borrowed is not None:   # a context owned by a Pool
            self.h = C.c_void_p(_borrowed)
            return
        h = C.c_void_p()
        rc = self.L.zk_ctx_create(device, C.byref(h))
        self.h = h
        self._chk(rc)

    def _chk(self, rc):
        if rc:
            detail = self.L.zk_last_error(self.h).decode()   # NULL handle: why zk_ctx_create failed (thread-local in the library)
            raise ZkError(rc, detail)

    def close(self):
        if getattr(self, 'h', None):
            if self._owned:
                self.L.zk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, nist_h64, tom_g72, tom_h72, sec_level=80):
        self._chk(self.L.zk_ctx_set_params(self.h, bytes(nist_h64), bytes(tom_g72), bytes(tom_h72), sec_level))
        self.sec = sec_level

    def set_ring(self, keys_be32, nkeys=None):
        keys_be32 = bytes(keys_be32)
        if nkeys is None:
            nkeys = len(keys_be32) // 32
        self._chk(self.L.zk_ctx_set_ring(self.h, keys_be32, nkeys))

    def keys_to_ints(self, pk_xy64):
        """keyToInt for n keys (64-byte affine each): (n x 32-byte big-endian ring entries, list of statuses)."""
        pk_xy64 = bytes(pk_xy64)
        n = len(pk_xy64) // 64
        out = C.create_string_buffer(32 * n)
        st = (C.c_int32 * n)()
        self._chk(self.L.zk_keys_to_ints(self.h, n, pk_xy64, out, st))
        return out.raw, list(st)

    def set_ring_device(self, dev_ptr, nkeys):
        self._chk(self.L.zk_ctx_set_ring_device(self.h, dev_ptr, nkeys))

    def set_chunk(self, chunk):
        self._chk(self.L.zk_ctx_set_chunk(self.h, chunk))

    def set_lanes(self, lanes):
        self._chk(self.L.zk_ctx_set_lanes(self.h, lanes))

    def set_mode(self, mode):
        """MODE_REFERENCE (default, byte parity with the reference) or MODE_HARDENED (statement hashed into the GK challenge)."""
        self._chk(self.L.zk_ctx_set_mode(self.h, int(mode)))

    def ring_digest(self):
        d = C.create_string_buffer(32)
        self._chk(self.L.zk_ring_digest(self.h, d))
        return d.raw

    def set_verify_groups(self, groups):
        self._chk(self.L.zk_ctx_set_verify_groups(self.h, groups))

    def set_ring_fold(self, matrix_pipe):
        self._chk(self.L.zk_ctx_set_ring_fold(self.h, 1 if matrix_pipe else 0))

    def set_wire(self, packed):
        """zk_ctx_set_wire: False / 0 = ZKA1, True / 1 = ZKA1P (33-byte Tom coordinates) for what the prover emits and the verifier is handed"""
        self._chk(self.L.zk_ctx_set_wire(self.h, 1 if packed else 0))

    def set_slice(self, proofs):
        """Proofs per PointAdd slice of the prover (0 = automatic: 4096 with page-locked output, none otherwise)."""
        self._chk(self.L.zk_ctx_set_slice(self.h, int(proofs)))

    def set_host_taper(self, on):
        """Tapered chunk plan of the host-pointer calls on page-locked buffers (default on)."""
        self._chk(self.L.zk_ctx_set_host_taper(self.h, 1 if on else 0))

    def set_comb_bits(self, bits):
        """Comb width of the Tom-256 fixed-base tables (8..24 unsigned, 25/26 signed digits); call before set_params."""
        self._chk(self.L.zk_ctx_set_comb_bits(self.h, bits))

    def wipe(self):
        """zero the witness-derived device memory of the context (prover workspaces, staged inputs); also done by close() and after a failed prove call"""
        self.L.zk_ctx_wipe.argtypes = [C.c_void_p]
        self.L.zk_ctx_wipe.restype = C.c_int
        self._chk(self.L.zk_ctx_wipe(self.h))

    def set_batch_verify(self, min_chunk):
        """Chunks of >= min_chunk proofs (default 256; 0 never, 1 always) get the chunk-wide bucket-method check of the
        Tom-256 relations first; the per-proof sums only run when it fails."""
        self._chk(self.L.zk_ctx_set_batch_verify(self.h, int(min_chunk)))

    def proof_max_size(self):
        return self.L.zk_proof_max_size(self.h)

    def prove_batch(self, msg, sig, pk, which, seeds=None, streams=None, stream_blocks=0):
        """Host-buffer entry point.  Returns (list of proof bytes or None, list of status).
        seeds: B x 32 bytes, FRESH, SECRET and distinct per proof (every blinding factor of proof b derives from seed b: a
        reused or guessable seed reveals the witness).  With neither seeds nor streams the seeds come from os.urandom."""
        B = len(which)
        if seeds is None and streams is None:
            seeds = os.urandom(32 * B)
        if streams is None and len(seeds) != 32 * B:
            raise ValueError('seeds must hold 32 bytes per proof')
        cap = self.proof_max_size() * max(B, 1)
        out = C.create_string_buffer(cap)
        off = (C.c_uint64 * (B + 1))()
        st = (C.c_int32 * B)()
        w = (C.c_uint32 * B)(*which)
        if streams is None:
            data = C.create_string_buffer(bytes(seeds), 32 * B)
            rng = ZkRng(0, C.cast(data, C.c_void_p), 0)
        else:
            data = C.create_string_buffer(bytes(streams), 32 * B * stream_blocks)
            rng = ZkRng(1, C.cast(data, C.c_void_p), stream_blocks)
        self._chk(self.L.zk_prove_batch(self.h, B, bytes(msg), bytes(sig), bytes(pk), w, C.byref(rng), out, cap, off, st))
        raw = out.raw
        proofs = [raw[off[b]:off[b + 1]] if st[b] == 0 else None for b in range(B)]
        return proofs, list(st)

    def prove_batch_host_raw(self, msg, sig, pk, which, seeds, out=None):
        """zk_prove_batch on host buffers without slicing the output.  out: a PinnedBuffer (overlapped DMA), a ctypes byte array
        (pageable) or None (a pageable array is allocated).  Returns (wall seconds of the C call, out buffer, offsets, statuses)."""
        import time
        B = len(which)
        cap = self.proof_max_size() * max(B, 1)
        if out is None:
            out = (C.c_uint8 * cap)()
        if isinstance(out, PinnedBuffer):
            cap, optr = min(cap, out.nbytes), out.ptr
        else:
            cap, optr = min(cap, C.sizeof(out)), C.addressof(out)
        off = (C.c_uint64 * (B + 1))()
        st = (C.c_int32 * B)()
        w = (C.c_uint32 * B)(*which)
        data = C.create_string_buffer(bytes(seeds), 32 * B)
        rng = ZkRng(0, C.cast(data, C.c_void_p), 0)
        msg, sig, pk = bytes(msg), bytes(sig), bytes(pk)
        t0 = time.time()
        self._chk(self.L.zk_prove_batch(self.h, B, msg, sig, pk, w, C.byref(rng), optr, cap, off, st))
        return time.time() - t0, out, off, st

    def verify_batch_host_raw(self, msg, proofs, off, B, vseeds=None):
        """zk_verify_batch on a packed host buffer (PinnedBuffer or ctypes array) with offsets `off`.  Returns (seconds, ok, status)."""
        import time
        ok = (C.c_uint8 * B)()
        st = (C.c_int32 * B)()
        ptr = proofs.ptr if isinstance(proofs, PinnedBuffer) else C.addressof(proofs)
        msg = bytes(msg)
        t0 = time.time()
        self._chk(self.L.zk_verify_batch(self.h, B, msg, ptr, off, bytes(vseeds) if vseeds is not None else None, ok, st))
        return time.time() - t0, ok, st

    # ---- streamed calls (zk_prove_submit / zk_prove_wait ...): several batches in flight, waits in submission order
    def prove_submit(self, msg, sig, pk, which, seeds, out):
        """Queues one batch; `out` is a PinnedBuffer.  Returns a ticket for prove_wait (it keeps every buffer of the job alive)."""
        B = len(which)
        t = {'B': B, 'out': out, 'off': (C.c_uint64 * (B + 1))(), 'st': (C.c_int32 * B)(), 'w': (C.c_uint32 * B)(*which),
             'data': C.create_string_buffer(bytes(seeds), 32 * B), 'msg': bytes(msg), 'sig': bytes(sig), 'pk': bytes(pk), 'job': C.c_void_p()}
        t['rng'] = ZkRng(0, C.cast(t['data'], C.c_void_p), 0)
        cap = min(self.proof_max_size() * max(B, 1), out.nbytes)
        self._chk(self.L.zk_prove_submit(self.h, B, t['msg'], t['sig'], t['pk'], t['w'], C.byref(t['rng']), out.ptr, cap, t['off'], t['st'], C.byref(t['job'])))
        return t

    def prove_wait(self, t):
        """-> (offsets, statuses) of the ticket's batch; the proofs are in the ticket's PinnedBuffer."""
        self._chk(self.L.zk_prove_wait(self.h, t['job']))
        return t['off'], t['st']

    def verify_submit(self, msg, proofs, off, B, vseeds=None):
        t = {'B': B, 'proofs': proofs, 'off': off, 'ok': (C.c_uint8 * B)(), 'st': (C.c_int32 * B)(), 'msg': bytes(msg),
             'seeds': bytes(vseeds) if vseeds is not None else None, 'job': C.c_void_p()}
        self._chk(self.L.zk_verify_submit(self.h, B, t['msg'], proofs.ptr, off, t['seeds'], t['ok'], t['st'], C.byref(t['job'])))
        return t

    def verify_wait(self, t):
        self._chk(self.L.zk_verify_wait(self.h, t['job']))
        return t['ok'], t['st']

    def copy_probe(self, lane=0, nbytes=256 << 20, numa_node=-1):
        """(d2h GB/s, h2d GB/s) of a page-locked copy on that lane's copy stream (zk_ctx_copy_probe); numa_node >= 0 binds the host buffer"""
        a, b = C.c_float(), C.c_float()
        self._chk(self.L.zk_ctx_copy_probe(self.h, lane, nbytes, numa_node, C.byref(a), C.byref(b)))
        return round(a.value, 1), round(b.value, 1)

    def test_counter(self, which=0):
        return int(self.L.zk_test_counter(self.h, which))

    def set_key_tables(self, on):
        self._chk(self.L.zk_ctx_set_key_tables(self.h, 1 if on else 0))

    def prove_submit_device(self, B, d_msg, d_sig, d_pk, d_which, d_seeds, d_out, out_cap, d_off, d_status, mode=0, stride_blocks=0):
        """zk_prove_submit_device: every pointer a device address; returns the ticket for prove_wait."""
        t = {'rng': ZkRng(mode, d_seeds, stride_blocks), 'job': C.c_void_p(), 'off': None, 'st': None}
        self._chk(self.L.zk_prove_submit_device(self.h, B, d_msg, d_sig, d_pk, d_which, C.byref(t['rng']), d_out, out_cap, d_off, d_status, C.byref(t['job'])))
        return t

    def prove_batch_device(self, B, d_msg, d_sig, d_pk, d_which, d_seeds, d_out, out_cap, d_off, d_status, mode=0, stride_blocks=0):
        rng = ZkRng(mode, d_seeds, stride_blocks)
        self._chk(self.L.zk_prove_batch_device(self.h, B, d_msg, d_sig, d_pk, d_which, C.byref(rng), d_out, out_cap, d_off, d_status))

    def verify_batch(self, msg, proofs, vseeds=None):
        B = len(proofs)
        off = (C.c_uint64 * (B + 1))()
        o = 0
        for b, p in enumerate(proofs):
            off[b] = o
            o += len(p)
        off[B] = o
        ok = (C.c_uint8 * B)()
        st = (C.c_int32 * B)()
        self._chk(self.L.zk_verify_batch(self.h, B, bytes(msg), b''.join(proofs), off, bytes(vseeds) if vseeds is not None else None, ok, st))
        return list(ok), list(st)

    def verify_batch_device(self, B, d_msg, d_proofs, d_off, d_vseeds, d_ok, d_status):
        self._chk(self.L.zk_verify_batch_device(self.h, B, d_msg, d_proofs, d_off, d_vseeds, d_ok, d_status))

    def synth_params(self, seed):
        a, b, c = C.create_string_buffer(64), C.create_string_buffer(72), C.create_string_buffer(72)
        self._chk(self.L.zk_synth_params(self.h, seed, a, b, c))
        return a.raw, b.raw, c.raw

    def synth_workload(self, seed, nkeys, B):
        ring = C.create_string_buffer(32 * nkeys)
        msg, sig, pk = C.create_string_buffer(32 * max(B, 1)), C.create_string_buffer(64 * max(B, 1)), C.create_string_buffer(64 * max(B, 1))
        which = (C.c_uint32 * max(B, 1))()
        seeds = C.create_string_buffer(32 * max(B, 1))
        self._chk(self.L.zk_synth_workload(self.h, seed, nkeys, B, ring, msg, sig, pk, which, seeds))
        return ring.raw, msg.raw[:32 * B], sig.raw[:64 * B], pk.raw[:64 * B], list(which)[:B], seeds.raw[:32 * B]

    def last_timing(self):
        total = C.c_float()
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        n = self.L.zk_last_timing(self.h, C.byref(total), names, ms, 32)
        return total.value, {names[i].decode(): ms[i] for i in range(min(n, 32))}

    def set_timing(self, mode):
        """zk_ctx_set_timing: 0 never, 1 per-family events in every blocking call, 2 (default) only in calls of more than 8 192 proofs"""
        self.L.zk_ctx_set_timing.argtypes = [C.c_void_p, C.c_int]
        self.L.zk_ctx_set_timing.restype = C.c_int
        self._chk(self.L.zk_ctx_set_timing(self.h, int(mode)))

    def last_wall_ms(self):
        """earliest start -> latest end of the last call's timed kernel families (last_timing()[0] is their sum, which overlapping families exceed)"""
        self.L.zk_last_wall_ms.argtypes = [C.c_void_p]
        self.L.zk_last_wall_ms.restype = C.c_float
        return float(self.L.zk_last_wall_ms(self.h))

    # ---- unit-test hooks
    def test_field_op(self, which, op, a_list, b_list):
        n = len(a_list)
        a = b''.join(x.to_bytes(40, 'big') for x in a_list)
        b = b''.join(x.to_bytes(40, 'big') for x in b_list)
        out = C.create_string_buffer(40 * n)
        self._chk(self.L.zk_test_field_op(self.h, which, op, n, a, b, out))
        return [int.from_bytes(out.raw[40 * i:40 * i + 40], 'big') for i in range(n)]

    def test_tom_commit(self, v_list, r_list):
        n = len(v_list)
        out = C.create_string_buffer(72 * n)
        self._chk(self.L.zk_test_tom_commit(self.h, n, b''.join(x.to_bytes(32, 'big') for x in v_list),
                                            b''.join(x.to_bytes(32, 'big') for x in r_list), out))
        return [out.raw[72 * i:72 * i + 72] for i in range(n)]

    def test_p256_fixed_mul(self, base_sel, k_list):
        n = len(k_list)
        out = C.create_string_buffer(64 * n)
        self._chk(self.L.zk_test_p256_fixed_mul(self.h, base_sel, n, b''.join(x.to_bytes(32, 'big') for x in k_list), out))
        return [out.raw[64 * i:64 * i + 64] for i in range(n)]

    def test_sha256(self, msgs):
        n, ln = len(msgs), len(msgs[0])
        assert all(len(m) == ln for m in msgs)
        out = C.create_string_buffer(32 * n)
        self._chk(self.L.zk_test_sha256(self.h, n, ln, b''.join(msgs), out))
        return [out.raw[32 * i:32 * i + 32] for i in range(n)]

    def test_rng_draws(self, B, first_k, n_k, seeds=None, streams=None, stream_blocks=0):
        if streams is None:
            data = C.create_string_buffer(bytes(seeds), 32 * B)
            rng = ZkRng(0, C.cast(data, C.c_void_p), 0)
        else:
            data = C.create_string_buffer(bytes(streams), 32 * B * stream_blocks)
            rng = ZkRng(1, C.cast(data, C.c_void_p), stream_blocks)
        out = C.create_string_buffer(32 * B * n_k)
        self._chk(self.L.zk_test_rng_draws(self.h, B, C.byref(rng), first_k, n_k, out))
        return [[out.raw[32 * (b * n_k + j):32 * (b * n_k + j) + 32] for j in range(n_k)] for b in range(B)]


class Pool:
    """Several GPUs of one node behind one handle (zk_pool, include/zkattest.h): the batch is split into contiguous shards, the
    ring is uploaded once and broadcast device-to-device (RCCL over xGMI, peer copies as fallback)."""

    def __init__(self, device_ids):
        self.L = lib()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = self.L.zk_pool_create(ids, len(device_ids), C.byref(h))
        self.h = h
        self._chk(rc)
        self.n = len(device_ids)

    def _chk(self, rc):
        if rc:
            raise ZkError(rc, self.L.zk_pool_last_error(self.h).decode())   # NULL handle: why zk_pool_create failed

    def close(self):
        if getattr(self, 'h', None):
            self.L.zk_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        """zk_pool_last_error: after set_ring also why RCCL was not used"""
        return self.L.zk_pool_last_error(self.h).decode()

    def rccl_library(self):
        """zk_pool_rccl_library: the file the nccl* entry points came from ('' while RCCL was never loaded)"""
        return self.L.zk_pool_rccl_library(self.h).decode()

    def engine(self, i):
        """Per-device context (settings only: chunk, lanes, comb width, host taper); owned by the pool."""
        return Engine(_borrowed=self.L.zk_pool_ctx(self.h, i))

    def numa_node(self, i):
        return self.L.zk_pool_numa_node(self.h, i)

    def shard_ms(self):
        ms = (C.c_float * self.n)()
        self.L.zk_pool_shard_ms(self.h, ms, self.n)
        return [round(float(x), 2) for x in ms]

    def shard(self, B, i):
        f, c = C.c_uint64(), C.c_uint64()
        self.L.zk_pool_shard(self.h, B, i, C.byref(f), C.byref(c))
        return f.value, c.value

    def set_params(self, nist_h64, tom_g72, tom_h72, sec_level=80):
        self._chk(self.L.zk_pool_set_params(self.h, bytes(nist_h64), bytes(tom_g72), bytes(tom_h72), sec_level))

    def set_ring(self, keys_be32, nkeys=None):
        keys_be32 = bytes(keys_be32)
        self._chk(self.L.zk_pool_set_ring(self.h, keys_be32, nkeys if nkeys is not None else len(keys_be32) // 32))
        return self.L.zk_pool_ring_transport(self.h).decode()

    def prove_batch_raw(self, msg, sig, pk, which, seeds, out, cap):
        """zk_pool_prove_batch into `out` (PinnedBuffer or ctypes array).  Returns (seconds, off, len, status)."""
        import time
        B = len(which)
        off, ln, st = (C.c_uint64 * B)(), (C.c_uint64 * B)(), (C.c_int32 * B)()
        w = (C.c_uint32 * B)(*which)
        data = C.create_string_buffer(bytes(seeds), 32 * B)
        rng = ZkRng(0, C.cast(data, C.c_void_p), 0)
        optr = out.ptr if isinstance(out, PinnedBuffer) else C.addressof(out)
        msg, sig, pk = bytes(msg), bytes(sig), bytes(pk)
        t0 = time.time()
        self._chk(self.L.zk_pool_prove_batch(self.h, B, msg, sig, pk, w, C.byref(rng), optr, cap, off, ln, st))
        return time.time() - t0, off, ln, st

    def prove_batch_device_out(self, msg, sig, pk, which, seeds, d_out, caps):
        """zk_pool_prove_batch_device: d_out = per-device buffers from device_alloc.  Returns (seconds, off, len, status)."""
        import time
        B = len(which)
        off, ln, st = (C.c_uint64 * B)(), (C.c_uint64 * B)(), (C.c_int32 * B)()
        w = (C.c_uint32 * B)(*which)
        data = C.create_string_buffer(bytes(seeds), 32 * B)
        rng = ZkRng(0, C.cast(data, C.c_void_p), 0)
        ptrs = (C.c_void_p * self.n)(*d_out)
        cp = (C.c_uint64 * self.n)(*caps)
        msg, sig, pk = bytes(msg), bytes(sig), bytes(pk)
        t0 = time.time()
        self._chk(self.L.zk_pool_prove_batch_device(self.h, B, msg, sig, pk, w, C.byref(rng), ptrs, cp, off, ln, st))
        return time.time() - t0, off, ln, st

    def device_alloc(self, i, nbytes):
        p = self.L.zk_pool_device_alloc(self.h, i, nbytes)
        if not p:
            raise MemoryError('zk_pool_device_alloc(%d, %d) failed' % (i, nbytes))
        return p

    def device_free(self, i, p):
        self.L.zk_pool_device_free(self.h, i, p)

    # ---- streamed pool calls: tickets keep every buffer of the job alive until its wait
    def prove_submit(self, msg, sig, pk, which, seeds, out, cap):
        B = len(which)
        t = {'B': B, 'out': out, 'off': (C.c_uint64 * B)(), 'ln': (C.c_uint64 * B)(), 'st': (C.c_int32 * B)(), 'w': (C.c_uint32 * B)(*which),
             'data': C.create_string_buffer(bytes(seeds), 32 * B), 'msg': bytes(msg), 'sig': bytes(sig), 'pk': bytes(pk), 'job': C.c_void_p()}
        t['rng'] = ZkRng(0, C.cast(t['data'], C.c_void_p), 0)
        self._chk(self.L.zk_pool_prove_submit(self.h, B, t['msg'], t['sig'], t['pk'], t['w'], C.byref(t['rng']), out.ptr, cap, t['off'], t['ln'], t['st'], C.byref(t['job'])))
        return t

    def prove_wait(self, t):
        self._chk(self.L.zk_pool_prove_wait(self.h, t['job']))
        return t['off'], t['ln'], t['st']

    def test_fail_submit(self, slot):
        """fault injection (tests): the next streamed submit of this pool fails at device slot `slot` after the earlier slots were submitted.  Only the test
        build of the library has the entry point (csrc/Makefile `testhooks`: lib/libzkattest_hip_testhooks.so, selected with ZKATTEST_LIB)."""
        if not hasattr(self.L, 'zk_test_pool_fail_next_submit'):
            raise RuntimeError('this build has no fault injection (load lib/libzkattest_hip_testhooks.so through ZKATTEST_LIB)')
        self.L.zk_test_pool_fail_next_submit.argtypes = [C.c_void_p, C.c_int]
        self.L.zk_test_pool_fail_next_submit.restype = None
        self.L.zk_test_pool_fail_next_submit(self.h, int(slot))

    def verify_submit(self, msg, proofs, off, ln, B, vseeds=None):
        t = {'B': B, 'proofs': proofs, 'off': off, 'ln': ln, 'ok': (C.c_uint8 * B)(), 'st': (C.c_int32 * B)(), 'msg': bytes(msg),
             'seeds': bytes(vseeds) if vseeds is not None else None, 'job': C.c_void_p()}
        self._chk(self.L.zk_pool_verify_submit(self.h, B, t['msg'], proofs.ptr, off, ln, t['seeds'], t['ok'], t['st'], C.byref(t['job'])))
        return t

    def verify_wait(self, t):
        self._chk(self.L.zk_pool_verify_wait(self.h, t['job']))
        return t['ok'], t['st']

    def prove_batch(self, msg, sig, pk, which, seeds=None):
        B = len(which)
        if seeds is None:
            seeds = os.urandom(32 * B)
        e = self.engine(0)
        cap = (e.proof_max_size() * (B // self.n + 1) + 256) * self.n
        out = (C.c_uint8 * cap)()
        _, off, ln, st = self.prove_batch_raw(msg, sig, pk, which, seeds, out, cap)
        raw = bytes(out)
        return [raw[off[b]:off[b] + ln[b]] if st[b] == 0 else None for b in range(B)], list(st)

    def verify_batch_raw(self, msg, proofs, off, ln, B, vseeds=None):
        import time
        ok, st = (C.c_uint8 * B)(), (C.c_int32 * B)()
        ptr = proofs.ptr if isinstance(proofs, PinnedBuffer) else C.addressof(proofs)
        msg = bytes(msg)
        t0 = time.time()
        self._chk(self.L.zk_pool_verify_batch(self.h, B, msg, ptr, off, ln, bytes(vseeds) if vseeds is not None else None, ok, st))
        return time.time() - t0, ok, st

    def verify_batch(self, msg, proofs, vseeds=None):
        B = len(proofs)
        off, ln = (C.c_uint64 * B)(), (C.c_uint64 * B)()
        o = 0
        for b, p in enumerate(proofs):
            off[b], ln[b] = o, len(p)
            o += len(p)
        blob = (C.c_uint8 * max(o, 1)).from_buffer_copy(b''.join(proofs) or b'\0')
        _, ok, st = self.verify_batch_raw(msg, blob, off, ln, B, vseeds)
        return list(ok), list(st)
