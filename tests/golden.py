"""Loader for tests/golden/entries.json (made by oracle/gen_golden.py from the reference)."""
import base64
import hashlib
import json
import os

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "entries.json")


def sha(b):
    return hashlib.sha256(b).hexdigest()


def load():
    with open(PATH) as f:
        return json.load(f)


def load_items():
    """[(method, comp, uncomp_size, crc32, flags, fixture_dict)]"""
    return [(e["method"], base64.b64decode(e["comp"]), e["uncomp_size"], e["crc32"], e["flags"], e)
            for e in load()["entries"]]
