"""Loaders for tests/golden/*.json (fixtures made by tests/golden/make_golden.py)."""
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

VERDICT_CODE = {"Ok": 0, "InnerProductError": 1, "WeightedInnerProdError": 2, "RangeProofError": 3, "Malformed": 255}


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def bp_cases():
    return load("bp_cases.json")


def bpplus_cases():
    return load("bpplus_cases.json")
