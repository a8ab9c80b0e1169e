"""Client side of the path: key generation, encryption, decryption (host, numpy).

Mirrors /root/reference/client/gen_secure_keyset.cpp (parameter sets + new_random_gate_bootstrapping_
secret_keyset), client/encrypt_image.cpp:65-85 (v = 2*pixel - 255 -> lweSymEncrypt(v/4096, 2^-15)) and
client/decrypt_image.cpp:37-63 (lweSymDecrypt(., 4096) -> signed -> argmax). None of this is on the
GPU hot path; it produces the inputs the hot path consumes (evaluation key, fresh ciphertexts) in
exactly the layouts include/redsec_hip.h documents.

Everything is vectorised; negacyclic products with the binary TRLWE key are done as two exact
float64 matrix products on 16-bit halves (sums stay below 2^26).
"""
import numpy as np

MSG_SPACE = 4096          # client/decrypt_image.cpp:37 msg_space
SECALPHA = 2.0 ** -15     # client/encrypt_image.cpp:10

PARAM_SETS = {
    # name: (n, N, k, bk_l, bk_Bgbit, ks_t, ks_basebit, ks_stdev, bk_stdev)
    "default128": (630, 1024, 1, 3, 7, 8, 2, 2.0 ** -15, 2.0 ** -25),
    # client/gen_secure_keyset.cpp:70-91
    "redsec_small_v2": (350, 1024, 1, 10, 3, 9, 3, 2.0 ** -25, 2.0 ** -30),
    # client/gen_secure_keyset.cpp:47-68, 28-45, 9-26: the sets the reference defines beside the one it ships
    "redsec_small": (500, 1024, 1, 3, 10, 18, 1, 2.0 ** -25, 2.0 ** -36),
    "redsec_medium": (3072, 4096, 1, 3, 10, 18, 1, 2.0 ** -40, 2.0 ** -45),
    "redsec_large": (6144, 8192, 1, 3, 10, 18, 1, 2.0 ** -41, 2.0 ** -46),
}


def modswitch_to_torus32(mu, msize):
    """TFHE modSwitchToTorus32 (BinOps_enc.cpp:137,184,190; encrypt_image.cpp:77)."""
    interv = ((1 << 63) // int(msize)) * 2
    v = ((np.asarray(mu, dtype=np.int64).astype(object) * interv) >> 32) & 0xFFFFFFFF
    return _wrap32(np.asarray(v, dtype=object))


def _wrap32(x):
    x = np.asarray(x)
    if x.dtype == object:
        x = np.array([int(v) & 0xFFFFFFFF for v in x.ravel()], dtype=np.uint64).reshape(x.shape)
    return (x.astype(np.uint64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def _gaussian32(rng, sigma, shape):
    """TFHE gaussian32 with message 0: dtot32(N(0, sigma))."""
    e = rng.normal(0.0, sigma, shape)
    frac = e - np.trunc(e)
    return (frac * 4294967296.0).astype(np.int64).astype(np.uint64).astype(np.uint32).view(np.int32)


def _uniform32(rng, shape):
    return rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32).view(np.int32)


def _negacyclic_matrix(s):
    """T with (a * s)[j] = sum_m a[m] T[m][j] in Z[X]/(X^N+1), s binary."""
    N = len(s)
    idx = (np.arange(N)[None, :] - np.arange(N)[:, None])  # j - m
    T = s[idx % N].astype(np.float64)
    T[idx < 0] *= -1.0
    return T


def _mul_by_binary_poly(A, T):
    """Rows of A (int32 torus) times the binary key polynomial, exact mod 2^32."""
    Au = A.view(np.uint32).astype(np.uint64)
    hi = (Au >> 16).astype(np.float64)
    lo = (Au & 0xFFFF).astype(np.float64)
    ph = (hi @ T).astype(np.int64)
    pl = (lo @ T).astype(np.int64)
    return _wrap32(((ph << 16) + pl).astype(np.uint64))


class SecretKeySet:
    """TFheGateBootstrappingSecretKeySet: lwe_key, tlwe_key and the cloud (evaluation) key."""

    def __init__(self, name="redsec_small_v2", seed=0, n=None):
        (n0, N, k, l, Bgbit, t, basebit, ks_stdev, bk_stdev) = PARAM_SETS[name]
        self.name = name
        self.n = int(n) if n is not None else n0
        self.N, self.k, self.l, self.Bgbit, self.t, self.basebit = N, k, l, Bgbit, t, basebit
        self.W = self.n + 1
        rng = np.random.default_rng(seed)
        n = self.n
        self.lwe_key = rng.integers(0, 2, n).astype(np.int32)
        self.tlwe_key = rng.integers(0, 2, N).astype(np.int32)
        # --- bootstrapping key: [n][2l][2][N] ---
        rows = n * 2 * l
        A = _uniform32(rng, (rows, N))
        E = _gaussian32(rng, bk_stdev, (rows, N))
        Bp = _wrap32(E.view(np.uint32).astype(np.uint64) + _mul_by_binary_poly(A, _negacyclic_matrix(self.tlwe_key)).view(np.uint32))
        bk = np.empty((n, 2 * l, 2, N), np.int32)
        bk[:, :, 0, :] = A.reshape(n, 2 * l, N)
        bk[:, :, 1, :] = Bp.reshape(n, 2 * l, N)
        # tGswAddMuIntH: row c*l + j gets s_i * 2^(32-(j+1)Bgbit) on component c, coefficient 0
        for c in range(2):
            for j in range(l):
                h = np.uint32(1 << (32 - (j + 1) * Bgbit))
                cur = bk[:, c * l + j, c, 0].view(np.uint32)
                bk[:, c * l + j, c, 0] = (cur + self.lwe_key.astype(np.uint32) * h).view(np.int32)
        self.bk = np.ascontiguousarray(bk)
        # --- keyswitch key: [N][t][base][n+1], value 0 is the trivial zero sample ---
        base = 1 << basebit
        ksk = np.zeros((N, t, base, n + 1), np.int32)
        Ak = _uniform32(rng, (N, t, base - 1, n))
        Ek = _gaussian32(rng, ks_stdev, (N, t, base - 1))
        dot = (Ak.view(np.uint32).astype(np.uint64) * self.lwe_key.astype(np.uint64)).sum(axis=-1)
        v = np.arange(1, base, dtype=np.uint64)[None, None, :]
        shift = np.array([32 - (j + 1) * basebit for j in range(t)], dtype=np.uint64)[None, :, None]
        mess = (self.tlwe_key.astype(np.uint64)[:, None, None] * v) << shift
        ksk[:, :, 1:, :n] = Ak
        ksk[:, :, 1:, n] = _wrap32(mess + Ek.view(np.uint32).astype(np.uint64) + dot)
        self.ksk = np.ascontiguousarray(ksk)

    # lweSymEncrypt on a batch of torus32 messages -> int32 [B][n+1]
    def encrypt_torus(self, mu, alpha=SECALPHA, seed=1):
        mu = np.asarray(mu).astype(np.int64).ravel()
        rng = np.random.default_rng(seed)
        B = mu.size
        out = np.empty((B, self.W), np.int32)
        a = _uniform32(rng, (B, self.n))
        e = _gaussian32(rng, alpha, (B,))
        dot = (a.view(np.uint32).astype(np.uint64) * self.lwe_key.astype(np.uint64)).sum(axis=-1)
        out[:, :self.n] = a
        out[:, self.n] = _wrap32(mu.astype(np.uint64) + e.view(np.uint32).astype(np.uint64) + dot)
        return out

    def encrypt_bits(self, bits, seed=1):
        """bootsSymEncrypt: +-1/8."""
        e8 = 1 << 29
        return self.encrypt_torus(np.where(np.asarray(bits) != 0, e8, -e8), SECALPHA, seed)

    def encrypt_image(self, pixels, seed=1, preprocess="sign"):
        """client/encrypt_image.cpp:76-77: ptxt = 2*pixel - 255, message ptxt/4096. preprocess="relu": the
        ReLU nets' own input map pixel/100 - 1 (nets/mnist/relu1024x1/main.cpp:203), which the reference's
        client tool does not know about."""
        px = np.asarray(pixels, dtype=np.int64).ravel()
        v = (px // 100 - 1) if preprocess == "relu" else 2 * px - 255
        return self.encrypt_torus(v * (1 << 20), SECALPHA, seed)

    def phase(self, ct):
        ct = np.asarray(ct, np.int32).reshape(-1, self.W)
        dot = (ct[:, :self.n].view(np.uint32).astype(np.uint64) * self.lwe_key.astype(np.uint64)).sum(axis=-1)
        return _wrap32(ct[:, self.n].view(np.uint32).astype(np.uint64) - dot)

    def decrypt_bits(self, ct):
        return (self.phase(ct) > 0).astype(np.int64)

    def decrypt_ints(self, ct, msize=MSG_SPACE):
        """client/decrypt_image.cpp:52-58: round the phase to multiples of 1/msize, signed."""
        ph = self.phase(ct).view(np.uint32).astype(np.uint64)
        interv = 1 << (32 - int(np.log2(msize)))
        m = ((ph + interv // 2) // interv) % msize
        m = m.astype(np.int64)
        return np.where(m > msize // 2, m - msize, m)

    def classify(self, logits_ct):
        """client/decrypt_image.cpp:61-62 argmax."""
        return int(np.argmax(self.decrypt_ints(logits_ct)))


# ---- TFHE v1.1 file formats (the reference's client/*.cpp and nets/*/*/main.cpp exchange these files) -------------
# Same layout as redsec_amd/host/tfhe_shim.cpp writes and reads (see the comment there; [TFHE-recalled]).
TFHE_UID = dict(lwe_sample=42, lwe_key=43, tlwe_key=45, tgsw_sample=47, ks_key=200, bk_key=201)


def _section(title, props):
    body = "".join("%s: %s\n" % (k, v) for k, v in sorted(props.items()))
    return ("-----BEGIN %s-----\n%s-----END %s-----\n" % (title, body, title)).encode()


def _read_section(f, want):
    head = f.readline().decode().rstrip("\n")
    assert head == "-----BEGIN %s-----" % want, (head, want)
    props = {}
    while True:
        line = f.readline().decode().rstrip("\n")
        if line.startswith("-----END "):
            return props
        k, v = line.split(": ", 1)
        props[k] = v


def write_tfhe_keyset(f, sk, secret, ks_stdev, bk_stdev, max_stdev=0.012467):
    """export_tfheGateBootstrapping{Secret,Cloud}KeySet_toFile for a SecretKeySet (binary file object)."""
    i32 = lambda v: np.int32(v).tobytes()
    f.write(_section("GATEBOOTSPARAMS", dict(ks_t=sk.t, ks_basebit=sk.basebit)))
    f.write(_section("LWEPARAMS", dict(n=sk.n, alpha_min=repr(float(ks_stdev)), alpha_max=repr(float(max_stdev)))))
    f.write(_section("TLWEPARAMS", dict(N=sk.N, k=sk.k, alpha_min=repr(float(bk_stdev)), alpha_max=repr(float(max_stdev)))))
    f.write(_section("TGSWPARAMS", dict(l=sk.l, Bgbit=sk.Bgbit)))
    f.write(i32(TFHE_UID["bk_key"]))
    f.write(_section("LWEKSPARAMS", dict(n=sk.k * sk.N, t=sk.t, basebit=sk.basebit)))
    f.write(i32(TFHE_UID["ks_key"]) + np.float64(ks_stdev ** 2).tobytes())
    f.write(np.ascontiguousarray(sk.ksk, np.int32).tobytes())
    for i in range(sk.n):
        f.write(i32(TFHE_UID["tgsw_sample"]) + np.float64(bk_stdev ** 2).tobytes())
        f.write(np.ascontiguousarray(sk.bk[i], np.int32).tobytes())
    if secret:
        f.write(i32(TFHE_UID["lwe_key"]) + np.ascontiguousarray(sk.lwe_key, np.int32).tobytes())
        f.write(i32(TFHE_UID["tlwe_key"]) + np.ascontiguousarray(sk.tlwe_key, np.int32).tobytes())


def read_tfhe_keyset(f, secret):
    """-> dict(params..., bk [n][2l][2][N], ksk [N][t][base][n+1], and for secret files lwe_key, tlwe_key, uids seen)."""
    g = _read_section(f, "GATEBOOTSPARAMS"); lw = _read_section(f, "LWEPARAMS")
    tl = _read_section(f, "TLWEPARAMS"); tg = _read_section(f, "TGSWPARAMS")
    n, N, k, l, Bgbit = int(lw["n"]), int(tl["N"]), int(tl["k"]), int(tg["l"]), int(tg["Bgbit"])
    t, basebit = int(g["ks_t"]), int(g["ks_basebit"])
    rd = lambda count: np.frombuffer(f.read(4 * count), np.int32)
    uids = [int(rd(1)[0])]
    ks = _read_section(f, "LWEKSPARAMS")
    assert (int(ks["n"]), int(ks["t"]), int(ks["basebit"])) == (k * N, t, basebit)
    uids.append(int(rd(1)[0])); f.read(8)
    ksk = rd(k * N * t * (1 << basebit) * (n + 1)).reshape(k * N, t, 1 << basebit, n + 1)
    bk = np.empty((n, (k + 1) * l, k + 1, N), np.int32)
    for i in range(n):
        uids.append(int(rd(1)[0])); f.read(8)
        bk[i] = rd((k + 1) * l * (k + 1) * N).reshape((k + 1) * l, k + 1, N)
    out = dict(n=n, N=N, k=k, l=l, Bgbit=Bgbit, t=t, basebit=basebit, ks_stdev=float(lw["alpha_min"]), bk_stdev=float(tl["alpha_min"]),
               bk=bk, ksk=ksk)
    if secret:
        uids.append(int(rd(1)[0])); out["lwe_key"] = rd(n).copy()
        uids.append(int(rd(1)[0])); out["tlwe_key"] = rd(k * N).copy()
    assert f.read(1) == b"", "trailing bytes in key file"
    out["uids"] = uids
    return out


def write_ciphertexts(f, ct):
    """export_gate_bootstrapping_ciphertext_toFile per row of ct [B][n+1]: uid 42, a[n], b, double variance."""
    ct = np.ascontiguousarray(ct, np.int32)
    for row in ct:
        f.write(np.int32(TFHE_UID["lwe_sample"]).tobytes() + row.tobytes() + np.float64(0.0).tobytes())


def read_ciphertexts(f, n, count):
    rec = 4 + 4 * (n + 1) + 8
    raw = f.read(rec * count)
    assert len(raw) == rec * count
    out = np.empty((count, n + 1), np.int32)
    for i in range(count):
        assert np.frombuffer(raw[i * rec:i * rec + 4], np.int32)[0] == TFHE_UID["lwe_sample"]
        out[i] = np.frombuffer(raw[i * rec + 4:i * rec + 4 + 4 * (n + 1)], np.int32)
    return out


def synthetic_key_words(seed, count, first=0, chunk=1 << 22):
    """Words [first, first + count) of the synthetic key rs_load_synthetic_keys(seed) generates on the device (csrc/rs_ntt.h,
    synthetic_key_word: the high half of splitmix64(seed + k)); the keyswitch key uses seed ^ 0x6b73."""
    out = np.empty(int(count), np.int32)
    seed = np.uint64(int(seed) & (2**64 - 1))
    with np.errstate(over="ignore"):
        for lo in range(0, int(count), chunk):
            hi = min(int(count), lo + chunk)
            k = np.arange(first + lo + 1, first + hi + 1, dtype=np.uint64)
            z = seed + k * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z ^= z >> np.uint64(31)
            out[lo:hi] = (z >> np.uint64(32)).astype(np.uint32).view(np.int32)
    return out
