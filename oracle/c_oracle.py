"""oracle/c_oracle.py — TEST INFRASTRUCTURE.  ctypes front-end of oracle/liborc.so (srlz_oracle.c) plus the composition
of its operators into the reference's auto-encoder training step (models/learner.py:373-489 for --losses autoencoder:
CNNAutoEncoder.forward on obs and next_obs, autoEncoderLoss, backward).  numpy in / numpy out, reference layouts.
Pinned by tests/test_c_oracle.py against torch (operator level) and the reference's golden fixtures (step level).
"""
import ctypes
import os
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborc.so")
_F = ctypes.POINTER(ctypes.c_float)
_I32 = ctypes.POINTER(ctypes.c_int32)
_I64 = ctypes.POINTER(ctypes.c_int64)


def _load():
    if not os.path.exists(_LIB):
        raise RuntimeError("oracle/liborc.so missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = ctypes.CDLL(_LIB)
    lib.orc_sqdiff_sum.restype = ctypes.c_double
    lib.orc_kl_sum.restype = ctypes.c_double
    lib.orc_cross_entropy.restype = ctypes.c_double
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_F)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def conv2d_fwd(x, w, b, stride, pad):
    x, w = _f(x), _f(w)
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    y = np.empty((N, K, (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1), np.float32)
    lib().orc_conv2d_fwd(_p(x), _p(w), _p(None if b is None else _f(b)), _p(y), N, C, H, W, K, R, S, stride, pad)
    return y


def conv2d_bwd(x, w, dy, stride, pad, need_dx=True, has_bias=False):
    x, w, dy = _f(x), _f(w), _f(dy)
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    dx = np.empty_like(x) if need_dx else None
    dw = np.empty_like(w)
    db = np.empty(K, np.float32) if has_bias else None
    lib().orc_conv2d_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), N, C, H, W, K, R, S, stride, pad)
    return dx, dw, db


def convT2d_fwd(x, w, b, stride, pad=0):
    x, w = _f(x), _f(w)
    N, C, H, W = x.shape
    _, K, R, S = w.shape
    y = np.empty((N, K, (H - 1) * stride - 2 * pad + R, (W - 1) * stride - 2 * pad + S), np.float32)
    lib().orc_convT2d_fwd(_p(x), _p(w), _p(None if b is None else _f(b)), _p(y), N, C, H, W, K, R, S, stride, pad)
    return y


def convT2d_bwd(x, w, dy, stride, pad=0):
    x, w, dy = _f(x), _f(w), _f(dy)
    N, C, H, W = x.shape
    _, K, R, S = w.shape
    dx, dw, db = np.empty_like(x), np.empty_like(w), np.empty(K, np.float32)
    lib().orc_convT2d_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), N, C, H, W, K, R, S, stride, pad)
    return dx, dw, db


def bn_train_fwd(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    x = _f(x)
    N, C = x.shape[:2]
    HW = int(np.prod(x.shape[2:]))
    y, mean, invstd = np.empty_like(x), np.empty(C, np.float32), np.empty(C, np.float32)
    lib().orc_bn_train_fwd(_p(x), _p(_f(gamma)), _p(_f(beta)), _p(running_mean), _p(running_var), _p(y), _p(mean),
                           _p(invstd), N, C, HW, ctypes.c_float(momentum), ctypes.c_float(eps))
    return y, mean, invstd


def bn_eval_fwd(x, gamma, beta, running_mean, running_var, eps=1e-5):
    x = _f(x)
    N, C = x.shape[:2]
    y = np.empty_like(x)
    lib().orc_bn_eval_fwd(_p(x), _p(_f(gamma)), _p(_f(beta)), _p(_f(running_mean)), _p(_f(running_var)), _p(y), N, C,
                          int(np.prod(x.shape[2:])), ctypes.c_float(eps))
    return y


def bn_train_bwd(x, dy, gamma, mean, invstd):
    x, dy = _f(x), _f(dy)
    N, C = x.shape[:2]
    dx, dg, db = np.empty_like(x), np.empty(C, np.float32), np.empty(C, np.float32)
    lib().orc_bn_train_bwd(_p(x), _p(dy), _p(_f(gamma)), _p(mean), _p(invstd), _p(dx), _p(dg), _p(db), N, C,
                           int(np.prod(x.shape[2:])))
    return dx, dg, db


def relu_fwd(x):
    x = _f(x)
    y = np.empty_like(x)
    lib().orc_relu_fwd(_p(x), _p(y), ctypes.c_size_t(x.size))
    return y


def relu_bwd(x, dy):
    x, dy = _f(x), _f(dy)
    dx = np.empty_like(x)
    lib().orc_relu_bwd(_p(x), _p(dy), _p(dx), ctypes.c_size_t(x.size))
    return dx


def maxpool_fwd(x, pad):
    x = _f(x)
    N, C, H, W = x.shape
    OH, OW = (H + 2 * pad - 3) // 2 + 1, (W + 2 * pad - 3) // 2 + 1
    y, idx = np.empty((N, C, OH, OW), np.float32), np.empty((N, C, OH, OW), np.int32)
    lib().orc_maxpool_fwd(_p(x), _p(y), idx.ctypes.data_as(_I32), N, C, H, W, pad)
    return y, idx


def maxpool_bwd(dy, idx, in_shape):
    dy = _f(dy)
    N, C, H, W = in_shape
    dx = np.empty(in_shape, np.float32)
    lib().orc_maxpool_bwd(_p(dy), idx.ctypes.data_as(_I32), _p(dx), N, C, H, W, dy.shape[2], dy.shape[3])
    return dx


def linear_fwd(x, w, b):
    x, w = _f(x), _f(w)
    M, K = x.shape
    N = w.shape[0]
    y = np.empty((M, N), np.float32)
    lib().orc_linear_fwd(_p(x), _p(w), _p(None if b is None else _f(b)), _p(y), M, N, K)
    return y


def linear_bwd(x, w, dy):
    x, w, dy = _f(x), _f(w), _f(dy)
    M, K = x.shape
    N = w.shape[0]
    dx, dw, db = np.empty_like(x), np.empty_like(w), np.empty(N, np.float32)
    lib().orc_linear_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), M, N, K)
    return dx, dw, db


def sqdiff_sum(a, b):
    a, b = _f(a), _f(b)
    return float(lib().orc_sqdiff_sum(_p(a), _p(b), ctypes.c_size_t(a.size)))


def kl_sum(mu, logvar):
    mu, logvar = _f(mu), _f(logvar)
    return float(lib().orc_kl_sum(_p(mu), _p(logvar), ctypes.c_size_t(mu.size)))


def cross_entropy(logits, target):
    logits = _f(logits)
    target = np.ascontiguousarray(target, dtype=np.int64)
    B, A = logits.shape
    dl = np.empty_like(logits)
    val = lib().orc_cross_entropy(_p(logits), target.ctypes.data_as(_I64), B, A, _p(dl))
    return float(val), dl


def adam_step(p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    lib().orc_adam_step(_p(p), _p(_f(g)), _p(m), _p(v), ctypes.c_size_t(p.size), ctypes.c_double(lr), ctypes.c_double(b1),
                        ctypes.c_double(b2), ctypes.c_double(eps), step)


# ---------------------------------------------------------------------------------------------------------------
# Composition: CNNAutoEncoder forward / backward (models/models.py:47-114, models/autoencoders.py:84-118)
# ---------------------------------------------------------------------------------------------------------------
ENC = ((0, 1, 2, 3, 1), (4, 5, 1, 1, 0), (8, 9, 2, 1, 0))  # conv idx, bn idx, stride, pad, pool pad
DEC = ((0, 1), (3, 4), (6, 7), (9, 10))


def ae_forward(sd, x, cache):
    """sd: {name: np.ndarray} with the reference's keys; BN running stats are updated in place (train mode)."""
    a = _f(x)
    for ci, bi, st, pad, ppad in ENC:
        p = "model.encoder_conv.%d" % ci
        q = "model.encoder_conv.%d" % bi
        y = conv2d_fwd(a, sd[p + ".weight"], None, st, pad)
        z, mean, invstd = bn_train_fwd(y, sd[q + ".weight"], sd[q + ".bias"], sd[q + ".running_mean"], sd[q + ".running_var"])
        sd[q + ".num_batches_tracked"] += 1
        r = relu_fwd(z)
        pooled, idx = maxpool_fwd(r, ppad)
        cache.append(("enc", ci, bi, st, pad, a, y, z, mean, invstd, r.shape, idx))
        a = pooled
    e = a.reshape(a.shape[0], -1)
    states = linear_fwd(e, sd["model.encoder_fc.0.weight"], sd["model.encoder_fc.0.bias"])
    d = linear_fwd(states, sd["model.decoder_fc.0.weight"], sd["model.decoder_fc.0.bias"])
    cache.append(("fc", e, states, a.shape))
    a = d.reshape(-1, 64, 6, 6)
    for ci, bi in DEC:
        p = "model.decoder_conv.%d" % ci
        q = "model.decoder_conv.%d" % bi
        y = convT2d_fwd(a, sd[p + ".weight"], sd[p + ".bias"], 2)
        z, mean, invstd = bn_train_fwd(y, sd[q + ".weight"], sd[q + ".bias"], sd[q + ".running_mean"], sd[q + ".running_var"])
        sd[q + ".num_batches_tracked"] += 1
        cache.append(("dec", ci, bi, a, y, z, mean, invstd))
        a = relu_fwd(z)
    out = convT2d_fwd(a, sd["model.decoder_conv.12.weight"], sd["model.decoder_conv.12.bias"], 2)
    cache.append(("out", a))
    return states, out


def ae_backward(sd, cache, dout, dstates, grads):
    def acc(name, g):
        grads[name] = g if name not in grads else grads[name] + g
    kind, a = cache.pop()
    da, dw, db = convT2d_bwd(a, sd["model.decoder_conv.12.weight"], dout, 2)
    acc("model.decoder_conv.12.weight", dw)
    acc("model.decoder_conv.12.bias", db)
    while cache[-1][0] == "dec":
        _, ci, bi, a, y, z, mean, invstd = cache.pop()
        q = "model.decoder_conv.%d" % bi
        dz = relu_bwd(z, da)
        dy, dg, dbt = bn_train_bwd(y, dz, sd[q + ".weight"], mean, invstd)
        acc(q + ".weight", dg)
        acc(q + ".bias", dbt)
        p = "model.decoder_conv.%d" % ci
        da, dw, db = convT2d_bwd(a, sd[p + ".weight"], dy, 2)
        acc(p + ".weight", dw)
        acc(p + ".bias", db)
    _, e, states, pshape = cache.pop()
    dd = da.reshape(da.shape[0], -1)
    dst, dw, db = linear_bwd(states, sd["model.decoder_fc.0.weight"], dd)
    acc("model.decoder_fc.0.weight", dw)
    acc("model.decoder_fc.0.bias", db)
    if dstates is not None:
        dst = dst + dstates
    de, dw, db = linear_bwd(e, sd["model.encoder_fc.0.weight"], dst)
    acc("model.encoder_fc.0.weight", dw)
    acc("model.encoder_fc.0.bias", db)
    da = de.reshape(pshape)
    while cache and cache[-1][0] == "enc":
        _, ci, bi, st, pad, a, y, z, mean, invstd, rshape, idx = cache.pop()
        q = "model.encoder_conv.%d" % bi
        dr = maxpool_bwd(da, idx, rshape)
        dz = relu_bwd(z, dr)
        dy, dg, dbt = bn_train_bwd(y, dz, sd[q + ".weight"], mean, invstd)
        acc(q + ".weight", dg)
        acc(q + ".bias", dbt)
        p = "model.encoder_conv.%d" % ci
        da, dw, _ = conv2d_bwd(a, sd[p + ".weight"], dy, st, pad, need_dx=(ci != 0))
        acc(p + ".weight", dw)


def ae_train_step(state_dict, obs, next_obs, weight=1.0):
    """loss = w * (mse(obs, dec) + mse(next_obs, next_dec)); returns dict(total, states, next_states, decoded,
    next_decoded, grads, sd) — the C-oracle counterpart of torch_twin.train_step for --losses autoencoder."""
    sd = OrderedDict((k, np.array(v, dtype=np.float32 if np.asarray(v).dtype.kind == "f" else np.asarray(v).dtype))
                     for k, v in state_dict.items())
    c1, c2 = [], []
    st, dec = ae_forward(sd, obs, c1)
    nst, ndec = ae_forward(sd, next_obs, c2)
    n = float(obs.size)
    loss = sqdiff_sum(obs, dec) / n + sqdiff_sum(next_obs, ndec) / n
    grads = {}
    ae_backward(sd, c2, (weight * 2.0 / n) * (ndec - _f(next_obs)), None, grads)
    ae_backward(sd, c1, (weight * 2.0 / n) * (dec - _f(obs)), None, grads)
    return dict(total=weight * loss, states=st, next_states=nst, decoded=dec, next_decoded=ndec, grads=grads, sd=sd)
