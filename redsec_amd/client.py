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

    def encrypt_image(self, pixels, seed=1):
        """client/encrypt_image.cpp:76-77: ptxt = 2*pixel - 255, message ptxt/4096."""
        v = 2 * np.asarray(pixels, dtype=np.int64).ravel() - 255
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
