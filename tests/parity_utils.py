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
