#!/usr/bin/env python3
"""Generate tests/golden/ckpt_tables.json from the reference's shipped checkpoint index files.

Run in the build container only (needs /root/reference).  The output is DATA (variable name, dtype, shape,
offset, size per checkpoint) -- the reference's own known-answer table for the parameter-naming contract
 and for epc-net_amd/tf_bundle.py.  Each ``.index`` file itself (4-16 KB binary table, a
data file the reference ships) is also copied to tests/golden/ so the reader is tested on the real bytes.
"""
import importlib.util
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/exp"
CKPTS = {
    "epc-net": "epc-net/saved_model/model_epoch22_iter18101.ckpt",
    "epc-net-l": "epc-net-l/saved_model/model_epoch13_iter18101.ckpt",
    "epc-net-l-d_student": "epc-net-l-d/saved_model/student_model_epoch20_iter18101.ckpt",
    "epc-net-l-d_teacher": "epc-net-l-d/transfer_teacher/teacher_model_epoch22_iter18101.ckpt",
}


def main():
    spec = importlib.util.spec_from_file_location("tf_bundle", os.path.join(ROOT, "epc-net_amd", "tf_bundle.py"))
    tfb = importlib.util.module_from_spec(spec)
    sys.modules["tf_bundle"] = tfb
    spec.loader.exec_module(tfb)
    out = {}
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    for tag, rel in CKPTS.items():
        idx = os.path.join(REF, rel + ".index")
        entries = tfb.read_index(idx)
        out[tag] = {
            "source": "exp/" + rel + ".index",
            "entries": [
                {"name": e.name, "dtype": e.dtype.name, "shape": list(e.shape), "offset": e.offset, "size": e.size}
                for e in entries.values()
            ],
        }
        shutil.copyfile(idx, os.path.join(gold, tag + ".ckpt.index"))
    with open(os.path.join(gold, "ckpt_tables.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for tag, v in out.items():
        print(tag, len(v["entries"]), "entries")


if __name__ == "__main__":
    main()
