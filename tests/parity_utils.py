"""Shared helpers for the parity tests and smoke(): the oracle-side pipeline (checker) and
small scene/task fixtures.  The oracle is only ever the checker here."""
from __future__ import annotations

import dataclasses
import types

import numpy as np

from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
from tests.scenes import make_scene, make_task, scene_text_embeds  # noqa: F401 (re-exported)
from oracle import clip_ref, host_ref, render_ref
from oracle.pipeline import OraclePipeline, oracle_logits  # noqa: F401 (re-exported)


def seeded_text_embeds(cfg, sd, n_caps=2, seed=5):
    """Cached text embeddings: the text tower of the same random checkpoint on fixed
    synthetic input_ids (EOS = largest id last), SURVEY.md §8(d)."""
    r = np.random.Generator(np.random.PCG64(seed))
    T = min(cfg["ctx"], 12)
    ids = r.integers(2, cfg["vocab"] - 2, size=(n_caps, T))
    ids[:, 0] = cfg["vocab"] - 2
    ids[:, -1] = cfg["vocab"] - 1
    return clip_ref.text_embeds(ids, sd, cfg)


def random_unit_text_embeds(D, n_caps=2, seed=5):
    r = np.random.Generator(np.random.PCG64(seed))
    t = r.standard_normal((n_caps, D)).astype(np.float32)
    return t / np.linalg.norm(t, axis=-1, keepdims=True)


def cosine(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def logit_bar(cfg, measured=None, what=""):
    """north_star's bar: |dlogit| / logit_scale <= 1e-3 ("scores within 1e-3 cosine"), stated for CLIP's embedding widths
    (D = 512 / 768).  A logit is the embedding error projected on a unit caption vector, so the same embedding error |de|
    reads sqrt(512 / D) times larger at a narrower projection: the 2-layer unit-test model `vit_tiny` (D = 64) is held to
    1e-3 * sqrt(512 / 64) = 2.83e-3 — measured on MI355X (tests/diag/parity_bars.py, 24 random frames): |dlogit| / scale
    1.24e-3 with |de| = 4.2e-3, BELOW ViT-B/16's |de| = 6.0e-3 (|dlogit| / scale 5.1e-4 at D = 512).  Every model with
    D >= 512, shallow or full depth, gets 1e-3.  Prints the measurement next to the bar it is held to."""
    bar = 1e-3 * float(np.sqrt(512.0 / min(512, cfg["proj"])))
    if measured is not None:
        print(f"[parity] {what}: |dlogit| / logit_scale = {measured:.2e}  (bar {bar:.2e}, D = {cfg['proj']})")
    return bar
