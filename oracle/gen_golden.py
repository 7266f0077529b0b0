#!/usr/bin/env python3
"""Generate tests/golden/ from the GENUINE reference (oracle/_ref/libjpegref.so).

Run in the authoring container (needs /root/reference):   python oracle/gen_golden.py
The reference itself cannot travel to the GPU box; these vectors (inputs + expected outputs,
plain data) do.  Nothing here is reference source text.

Fixtures written:
  tests/golden/images/*.jpg      the reference's own sample files (data files, images/)
  tests/golden/<name>.npz        per baseline image: desc, qtabs, coefficient blocks after
                                 decodeHuffman() (int16, decode order), RGB after
                                 YCbCrToRGB() (cropped), sha256 of the RGB bytes
  tests/golden/kat_blocks.npz    known-answer vectors through the reference hot path on
                                 synthetic blocks: DC-only, single-coefficient impulses,
                                 max-magnitude, truncation-stress and full-range random blocks
                                 in 4:4:4 / 4:2:2 / 4:4:0 / 4:2:0 with ragged sizes
  tests/golden/manifest.json     sizes + hashes + reference timings measured here
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Ref, make_desc  # noqa: E402
from jpeg_decoder_amd import synth  # noqa: E402

REF_IMAGES = "/root/reference/images"
GOLD = os.path.join(ROOT, "tests", "golden")
BASELINE = ["img", "img2", "img3", "img4", "img5", "img6"]


def kat_cases():
    """-> list of (name, width, height, hs, vs, qtab_id, coef, qtabs)."""
    q = synth.annex_k_qtabs(90)
    q[2] = np.arange(1, 65, dtype=np.uint16)   # a third, distinct table
    q[3] = 255                                 # the largest table the reference can hold
    cases = []
    layouts = [(1, 1), (2, 1), (1, 2), (2, 2)]
    for hs, vs in layouts:
        tag = f"{hs}x{vs}"
        for (w, h) in [(8, 8), (17, 9), (33, 47), (64, 40)]:
            _, _, bpm, n = synth.geometry(w, h, hs, vs)
            cases.append((f"rand_full_{tag}_{w}x{h}", w, h, hs, vs, (0, 1, 2),
                          synth.random_blocks(n, 1000 + w + h + hs * 7 + vs), q))
            cases.append((f"rand_small_{tag}_{w}x{h}", w, h, hs, vs, (0, 1, 1),
                          synth.random_blocks(n, 2000 + w + h + hs * 7 + vs, -255, 255), q))
            c, _ = synth.synth_blocks(w, h, hs, vs, image_index=w + h, qtabs=q, qtab_id=(0, 1, 2))
            cases.append((f"stat_{tag}_{w}x{h}", w, h, hs, vs, (0, 1, 2), c, q))
        # q = 255 everywhere with extreme coefficients: the largest magnitudes the reference
        # can produce without signed overflow
        _, _, bpm, n = synth.geometry(24, 16, hs, vs)
        ext = synth.random_blocks(n, 77 + hs + vs)
        ext[::3] = 32767
        ext[1::3] = -32768
        cases.append((f"extreme_{tag}", 24, 16, hs, vs, (3, 3, 3), ext, q))
    # impulse / DC-only blocks, 4:4:4, one MCU per impulse position (64 MCUs in a row)
    n = 64 * 3
    imp = np.zeros((n, 64), np.int16)
    for k in range(64):
        imp[3 * k + 0, k] = 37          # luma impulse at natural index k
        imp[3 * k + 1, k] = -19         # Cb
        imp[3 * k + 2, 63 - k] = 23     # Cr
    cases.append(("impulse_1x1", 512, 8, 1, 1, (0, 1, 1), imp, q))
    dc = np.zeros((n, 64), np.int16)
    dc[:, 0] = np.arange(n) * 11 - 1000
    cases.append(("dc_only_1x1", 512, 8, 1, 1, (0, 1, 1), dc, q))
    return cases


def main():
    ref = Ref()
    os.makedirs(os.path.join(GOLD, "images"), exist_ok=True)
    manifest = {"constants_f32_bits": [int(x) for x in ref.constants()], "images": {}, "kat": {}}
    for name in BASELINE + ["prograssive-sample-2"]:
        shutil.copyfile(os.path.join(REF_IMAGES, name + ".jpg"), os.path.join(GOLD, "images", name + ".jpg"))
        os.chmod(os.path.join(GOLD, "images", name + ".jpg"), 0o644)
    for name in BASELINE:
        path = os.path.join(REF_IMAGES, name + ".jpg")
        info, coef, q, rgb = ref.decode_file(path)
        sha = hashlib.sha256(rgb.tobytes()).hexdigest()
        desc = np.array([info.width, info.height, info.hs, info.vs, *info.qtab_id, info.restart_interval,
                         info.mcu_w_real, info.mcu_h_real], np.int32)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), desc=desc, qtabs=q, coef=coef, rgb=rgb)
        manifest["images"][name] = {
            "file_sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(),
            "width": info.width, "height": info.height, "hs": info.hs, "vs": info.vs,
            "qtab_id": list(info.qtab_id), "restart_interval": info.restart_interval,
            "mcu_w_real": info.mcu_w_real, "mcu_h_real": info.mcu_h_real,
            "n_coded_blocks": info.n_coded_blocks, "coef_range": [info.coef_min, info.coef_max],
            "rgb_sha256": sha,
            "ref_ms": {"huffman": round(info.ms_huffman, 3), "dequant": round(info.ms_dequant, 3),
                       "idct": round(info.ms_idct, 3), "colour": round(info.ms_colour, 3)},
        }
        print(name, info.width, info.height, sha[:16])
    kat = {}
    for (name, w, h, hs, vs, qid, coef, q) in kat_cases():
        d = make_desc(w, h, hs, vs, qid)
        rgb = ref.blocks_to_rgb(d, coef, q)
        kat[name + "__desc"] = np.array([w, h, hs, vs, *qid], np.int32)
        kat[name + "__qtabs"] = q
        kat[name + "__coef"] = coef
        kat[name + "__rgb"] = rgb
        manifest["kat"][name] = hashlib.sha256(rgb.tobytes()).hexdigest()
    np.savez_compressed(os.path.join(GOLD, "kat_blocks.npz"), **kat)
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("kat cases:", len(manifest["kat"]))


if __name__ == "__main__":
    main()
