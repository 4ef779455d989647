"""Shared helpers: load the committed golden fixtures and rebuild their inputs."""
import hashlib
import json
from pathlib import Path

import numpy as np

import corpus

GOLDEN = Path(__file__).resolve().parent / "golden"


def load(name):
    return json.loads((GOLDEN / name).read_text())


def make_input(spec) -> np.ndarray:
    kind = spec["kind"]
    if kind == "literal":
        return np.frombuffer(spec["text"].encode("latin-1"), dtype=np.uint8).copy()
    if kind == "small_alphabet":
        return corpus.small_alphabet(spec["seed"], spec["n"], spec["alphabet"].encode("latin-1"),
                                     spec.get("terminate", False))
    if kind == "text_block":
        a = corpus.text_block(spec["seed"], spec["index"], spec["n"],
                              needle=spec.get("needle", "Sherlock").encode("latin-1"),
                              needle_rate=spec.get("needle_rate", 2.76e-6))
        if spec.get("unterminated"):
            a = a[:-1].copy()
        for pos, s in spec.get("splice", []):
            b = s.encode("latin-1")
            p = pos if pos >= 0 else len(a) + pos
            a[p:p + len(b)] = np.frombuffer(b, dtype=np.uint8)
        return a
    raise ValueError(kind)


def generated_cases():
    """Yields (case_name, data, expect_entry) with the input's sha256 verified."""
    doc = load("ref_generated_vectors.json")
    for c in doc["cases"]:
        data = make_input(c["input"])
        assert data.size == c["len"], c["name"]
        assert hashlib.sha256(data.tobytes()).hexdigest() == c["sha256"], f"generator drift in {c['name']}"
        for e in c["expect"]:
            yield c["name"], data, e


def generated_icase_cases():
    """ignore_case vectors: expected values from the wrapper loops on the reference's simd::strcasestr
    (tests/golden/gen_golden.py: icase_cases)."""
    doc = load("ref_generated_vectors.json")
    for c in doc["icase_cases"]:
        data = make_input(c["input"])
        assert data.size == c["len"], c["name"]
        assert hashlib.sha256(data.tobytes()).hexdigest() == c["sha256"], f"generator drift in {c['name']}"
        for e in c["expect"]:
            yield c["name"], data, e
