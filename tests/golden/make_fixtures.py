#!/usr/bin/env python3
"""Regenerate the golden DATA fixtures from the reference's own test data.

Run in the build container only (needs /root/reference):  python tests/golden/make_fixtures.py

* fish.jpg, edges.jpg, linesDark.jpg, linesBright.jpg -- the JPEG byte arrays the
  reference's only test embeds as `xxd -i` headers (reference test/test.cpp:41-44, fixture
  headers test/*.h), written back as the binary files they were made from.  Data, not code.
* fish_u8.npy and the three *_u8.npy -- the same four images decoded to 8-bit gray with
  Pillow (stand-in for cv::imdecode(IMREAD_GRAYSCALE), test.cpp:53-56), so GPU-box tests
  do not depend on a JPEG decoder.
* taps_ref.json is produced separately by `make -C oracle -f ref_taps.mk golden`.
"""
import io, os, re, sys
import numpy as np
from PIL import Image

REF = os.environ.get("CVSTEER_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = {
    "fish": "Pterois_volitans_Manado-e_edit_smallest.h",
    "edges": "edges.h",
    "linesDark": "linesDark.h",
    "linesBright": "linesBright.h",
}

def xxd_bytes(path):
    text = open(path).read()
    body = text[text.index("{") + 1:text.index("}")]
    data = bytes(int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]{2}", body))
    m = re.search(r"_len\s*=\s*(\d+)", text)
    assert m and int(m.group(1)) == len(data), (path, len(data))
    return data

def main():
    for name, hdr in FIXTURES.items():
        raw = xxd_bytes(os.path.join(REF, "test", hdr))
        with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
            f.write(raw)
        img = np.asarray(Image.open(io.BytesIO(raw)).convert("L"), dtype=np.uint8)
        np.save(os.path.join(HERE, name + "_u8.npy"), img)
        print(name, len(raw), "bytes ->", img.shape, img.dtype)

if __name__ == "__main__":
    sys.exit(main())
