"""numpy restatements of the collector kernels (kernels_collect.hip), used only by the tests."""
import numpy as np

from util import rng_draw

SAMPLE_STREAM = 0x73616D70


def sample_uniforms(seed: int, batch: int, counter: int, num_actions: int) -> np.ndarray:
    """u[e, a] exactly as qg_sample_actions builds it (f32)."""
    env = np.arange(batch, dtype=np.uint64)
    base = rng_draw((seed ^ SAMPLE_STREAM) & (2**64 - 1), env, counter)
    hi = (base >> np.uint64(32)).astype(np.uint32)[:, None]
    lo = (base & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None]
    a = np.arange(num_actions, dtype=np.uint32)[None, :]
    with np.errstate(over="ignore"):
        x = hi + a * np.uint32(0x9E3779B9)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= lo
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return ((x >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)


def race_keys(logits: np.ndarray, u: np.ndarray, mask=None) -> np.ndarray:
    """log of the exponential-race times, f64: the sampled action is the argmin."""
    keys = np.log(-np.log(u.astype(np.float64))) - logits.astype(np.float64)
    if mask is not None:
        keys = np.where(mask.astype(bool), keys, np.inf)
    return keys


def log_softmax(logits: np.ndarray, mask=None) -> np.ndarray:
    x = logits.astype(np.float64)
    if mask is not None:
        x = np.where(mask.astype(bool), x, -np.inf)
    m = x.max(axis=1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=1, keepdims=True))


def gae_f32(rewards, values, dones, last_values, gamma, lam):
    """Same f32 operation order as gae_kernel (no FMA)."""
    T, B = rewards.shape
    f = np.float32
    g, gl = f(gamma), f(gamma) * f(lam)
    adv = np.zeros((T, B), dtype=f)
    ret = np.zeros((T, B), dtype=f)
    next_v = np.zeros(B, dtype=f) if last_values is None else last_values.astype(f)
    acc = np.zeros(B, dtype=f)
    for t in range(T - 1, -1, -1):
        nd = np.where(dones[t] != 0, f(0), f(1)).astype(f)
        v = values[t].astype(f)
        delta = (rewards[t].astype(f) + (g * next_v) * nd) - v
        acc = delta + (gl * nd) * acc
        adv[t] = acc
        ret[t] = acc + v
        next_v = v
    return adv, ret
