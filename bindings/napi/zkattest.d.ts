// Type declarations of bindings/napi/zkattest.js: the reference's public surface (src/index.ts:17-19) over the MI355X engine.
/// <reference types="node" />
export class Group { readonly name: string; readonly p: bigint; readonly order: bigint; eq(g: Group): boolean; generator(): Point; isOnGroup(p: Point): boolean }
export class Point { readonly group: Group; x: bigint; y: bigint; eq(o: Point): boolean; add(o: Point): Point; mul(k: bigint | Scalar): Point; isIdentity(): boolean }
export class Scalar { readonly group: Group; k: bigint; eq(o: Scalar): boolean }
export const p256: Group, tomEdwards256: Group, ALL_GROUPS: Group[]
export class PedersenParams { c: Group; g: Point; h: Point; constructor(c: Group, g: Point, h: Point); eq(o: PedersenParams): boolean }
export function generatePedersenParams(c: Group, g?: Point): PedersenParams
export class SystemParametersList {
    NistGroup: PedersenParams; ProofGroup: PedersenParams; SecLevel: number
    constructor(NistGroup: PedersenParams, ProofGroup: PedersenParams, SecLevel: number)
    eq(o: SystemParametersList): boolean
}
export interface MultProof { C_4: Point; A_x: Point; A_y: Point; A_z: Point; A_4_1: Point; A_4_2: Point; t_x: Scalar; t_y: Scalar; t_z: Scalar; t_rx: Scalar; t_ry: Scalar; t_rz: Scalar; t_r4: Scalar }
export interface EqualityProof { A_1: Point; A_2: Point; t_x: Scalar; t_r1: Scalar; t_r2: Scalar }
export interface PointAddProof { C_8: Point; C_10: Point; C_11: Point; C_13: Point; pi_8: MultProof; pi_10: MultProof; pi_11: MultProof; pi_13: MultProof; pi_x: EqualityProof; pi_y: EqualityProof }
export interface ExpProof { A: Point; Tx: Point; Ty: Point; alpha?: Scalar; beta1?: Scalar; beta2?: Scalar; beta3?: Scalar; z?: Scalar; z2?: Scalar; proof?: PointAddProof; r1?: Scalar; r2?: Scalar }
export interface GKProof { cl: Point[]; ca: Point[]; cb: Point[]; cd: Point[]; f: Scalar[]; za: Scalar[]; zb: Scalar[]; zd: Scalar }
export class SignatureProofList {
    /** the engine's binary form of the proof (ZKA1, include/zkattest.h); the members below are materialised from it on demand */
    readonly bytes: Buffer
    constructor(zka1: Buffer)
    readonly R: Point; readonly comS1: Point; readonly keyXcom: Point; readonly keyYcom: Point
    readonly expProof: ExpProof[]; readonly membershipProof: GKProof
    eq(o: SignatureProofList): boolean
}
export type PublicKey = Buffer | Uint8Array | import('crypto').KeyObject | CryptoKey
export function generateParamsList(secLevel?: number): SystemParametersList
/** hardened mode: NUMS generators + statement-bound membership challenge; not byte-compatible with the reference */
export function generateParamsListHardened(secLevel?: number, tag?: Uint8Array): SystemParametersList & { hardened: boolean }
export function keyToInt(publicKey: PublicKey): Promise<bigint>
export function proveSignatureList(params: SystemParametersList, msgHash: Uint8Array, sigBytes: Uint8Array, publicKey: PublicKey, which: number, keys: bigint[]): Promise<SignatureProofList>
export function verifySignatureList(params: SystemParametersList, msgHash: Uint8Array, keys: bigint[], proof: SignatureProofList): Promise<boolean>
export function proveSignatureListBatch(params: SystemParametersList, msgHashes: Uint8Array[], sigs: Uint8Array[], publicKeys: PublicKey[], whichs: number[], keys: bigint[] | Buffer): Promise<SignatureProofList[]>
/** booleans per proof; `errors[b]` holds what verifySignatureList would have thrown for proof b (null otherwise) -- a malformed proof never affects its neighbours */
export type Verdicts = boolean[] & { readonly errors: (Error | null)[] }
export function verifySignatureListBatch(params: SystemParametersList, msgHashes: Uint8Array[], keys: bigint[] | Buffer, proofs: (SignatureProofList | Buffer)[]): Promise<Verdicts>
type Newable<T> = new (...args: any[]) => T
export function writeJson<T>(type: Newable<T>, object: T): string
export function readJson<T>(type: Newable<T>, text: string): T
/** whole batches on every host core, off the event loop; null where an item is malformed */
export function writeJsonBatch(proofs: (SignatureProofList | Buffer)[], threads?: number): Promise<(string | null)[]>
export function readJsonBatch(texts: (string | Buffer)[], threads?: number): Promise<(SignatureProofList | null)[]>
/** closes every cached GPU context (they are keyed by SystemParametersList content and device list) */
export function shutdown(): void
export interface EngineParams { nistH: Buffer; tomG: Buffer; tomH: Buffer; secLevel?: number }
export class Engine {
    constructor(devices?: number | number[])
    close(): void
    info(): { devices: number; ringTransport: string; proofMaxSize: number }
    setOption(name: 'chunk' | 'lanes' | 'combBits' | 'hostTaper' | 'batchVerify' | 'mode' | 'slice' | 'ringFold' | 'verifyGroups' | 'wire' | 'inflight', value: number): void
    /** zero the witness-derived device memory (prover workspaces, staged signatures and seeds) of every device now; close() and a failed prove do it by themselves */
    wipe(): void
    setParams(p: EngineParams): void
    setRing(keys: Buffer | bigint[]): string
    keysToInts(pkxy: Buffer): { keys: Buffer; status: Buffer }
    proveBatch(msg: Buffer, sig: Buffer, pk: Buffer, which: number[] | Buffer, seeds?: Buffer): Buffer[]
    verifyBatch(msg: Buffer, proofs: Buffer[], seeds?: Buffer): Verdicts
    proveBatchAsync(msg: Buffer, sig: Buffer, pk: Buffer, which: number[] | Buffer, seeds?: Buffer): Promise<Buffer[]>
    /** Streamed form (zk_pool_prove_submit / _wait): up to `inflight` batches inside the engine; `out` is a page-locked Buffer (Engine.hostAlloc) owned by the job until its Promise settles. */
    static hostAlloc(bytes: number): Buffer
    proveStream(msg: Buffer, sig: Buffer, pk: Buffer, which: number[] | Buffer, seeds: Buffer | undefined, out: Buffer): Promise<{ proofs: Buffer[]; status: Int32Array; used: number }>
    verifyStream(msg: Buffer, blob: Buffer, offsets: Buffer, lengths: Buffer, seeds?: Buffer): Promise<{ ok: boolean[]; status: Int32Array }>
    verifyBatchAsync(msg: Buffer, proofs: Buffer[], seeds?: Buffer): Promise<Verdicts>
}
/** Wire layout of the proofs proveSignatureList / proveSignatureListBatch return: 'zka1' (default) or 'zka1p' (33-byte Tom coordinates, 5.3 % fewer bytes; env ZKATTEST_WIRE).
 *  verifySignatureList / ...Batch accept either layout per proof; writeJson gives the same text for both. */
export function setWireLayout(name: 'zka1' | 'zka1p'): void
export function getWireLayout(): 'zka1' | 'zka1p'
