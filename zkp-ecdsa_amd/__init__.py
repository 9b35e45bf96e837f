"""zkp-ecdsa_amd: MI355X-native engine for the ZKAttest proving path of cloudflare/zkp-ecdsa.
Importable as `zkp_ecdsa_amd` through the shim module at the repo root."""
from ._native import Engine, Pool, PinnedBuffer, hardened_h, MODE_REFERENCE, MODE_HARDENED, write_json, read_json, write_json_batch, read_json_batch, pack_proof, unpack_proof, ZkError, ZkRng, LIB_PATH, SYMBOLS, STATUS_TEXT, build, lib  # noqa: F401
