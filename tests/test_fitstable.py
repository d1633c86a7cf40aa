"""The FITS spectral table of the reference's Python layer (ref python/sim5fitstable.py) written without astropy:
conformance to the FITS standard checked by an independent parser written here, the resume semantics of the class, and
the reader of the module.  (Parity with astropy's byte layout is unpinned: no FITS library exists in this image.)"""
import os
import struct

import numpy as np
import pytest

from sim5_amd.fitstable import Sim5_FitsTable, params_crc, read_fits


def _cards(block):
    return [block[i:i + 80].decode("ascii") for i in range(0, len(block), 80)]


def _hdus(raw):
    """independent of the module: walk the 2880-byte blocks, return [(dict of cards, data bytes)]"""
    assert len(raw) % 2880 == 0
    pos, out = 0, []
    while pos < len(raw):
        hdr = {}
        end = False
        while not end:
            for c in _cards(raw[pos:pos + 2880]):
                assert len(c) == 80 and all(32 <= ord(ch) < 127 for ch in c)
                if c.startswith("END"):
                    assert c.strip() == "END"
                    end = True
                elif end:
                    assert c.strip() == ""                     # blank fill after END
                elif c[8:10] == "= ":
                    hdr[c[:8].strip()] = c[10:].split(" / ")[0].strip()
            pos += 2880
        size = 0
        if "XTENSION" in hdr:
            size = int(hdr["NAXIS1"]) * int(hdr["NAXIS2"]) + int(hdr["PCOUNT"])
        data = raw[pos:pos + size]
        pad = raw[pos + size:pos + size + (-size % 2880)]
        assert set(pad) <= {0}
        pos += size + (-size % 2880)
        out.append((hdr, data))
    return out


def test_file_conforms_to_the_fits_standard(tmp_path):
    fn = str(tmp_path / "t.fits")
    energies = 10.0 ** np.linspace(-1, 1, 7)
    params = [("spin", [0.0, 0.5, 0.9]), ("incl", [10.0, 60.0])]
    t = Sim5_FitsTable(fn, 10.0, 1e4, params, energies)
    assert t.total_grid_size == 6
    seen = []
    for index, gi, gv in t.generator():
        seen.append((index, tuple(gi), tuple(gv)))
        t.write(index, 0.1 * (index + 1), energies * (index + 1), energies * 0.5 * (index + 1))
    assert [s[0] for s in seen] == list(range(6))
    assert seen[1] == (1, (0, 1), (0.0, 60.0)) and seen[4] == (4, (2, 0), (0.9, 10.0))      # the last grid changes fastest
    t.save()
    raw = open(fn, "rb").read()
    hd = _hdus(raw)
    assert len(hd) == 3
    p, meta, spec = hd
    assert p[0]["SIMPLE"] == "T" and p[0]["NAXIS"] == "0" and p[0]["EXTEND"] == "T"
    assert p[0]["CRC"].strip("' ") == params_crc(10.0, 1e4, params, energies)
    # META: 16A + 1J + 1PE descriptor = 28 bytes per row, 3 + 2 rows, the grids on the heap as big-endian float32
    assert meta[0]["XTENSION"].strip("' ") == "BINTABLE" and meta[0]["EXTNAME"].strip("' ") == "META"
    assert (meta[0]["NAXIS1"], meta[0]["NAXIS2"], meta[0]["TFIELDS"]) == ("28", "5", "3")
    assert [meta[0]["TFORM%d" % i].strip("' ") for i in (1, 2, 3)] == ["16A", "1J", "1PE(7)"]
    rows = meta[1][:28 * 5]
    heap = meta[1][28 * 5:]
    names, grids = [], []
    for i in range(5):
        r = rows[28 * i:28 * (i + 1)]
        n, cnt, off = struct.unpack(">iii", r[16:28])
        assert n == cnt
        names.append(r[:16].decode().strip())
        grids.append(np.frombuffer(heap[off:off + 4 * cnt], ">f4"))
    assert names == ["REF_MASS", "REF_DIST", "ENERGIES", "SPIN", "INCL"]
    assert grids[0][0] == 10.0 and grids[1][0] == 1e4 and np.allclose(grids[2], energies, rtol=1e-7) and grids[4].tolist() == [10.0, 60.0]
    assert int(meta[0]["PCOUNT"]) == len(heap) == 4 * (1 + 1 + 7 + 3 + 2)
    # SPECTRA: 1E + 7E + 7E
    assert spec[0]["EXTNAME"].strip("' ") == "SPECTRA" and (spec[0]["NAXIS1"], spec[0]["NAXIS2"]) == (str(4 + 56), "6")
    assert [spec[0]["TFORM%d" % i].strip("' ") for i in (1, 2, 3)] == ["1E", "7E", "7E"]
    row3 = np.frombuffer(spec[1][60 * 3:60 * 4], ">f4")
    assert np.isclose(row3[0], 0.4) and np.allclose(row3[1:8], energies * 4, rtol=1e-6) and np.allclose(row3[8:], energies * 2, rtol=1e-6)


def test_resume_and_metadata_check(tmp_path):
    fn = str(tmp_path / "r.fits")
    energies = [0.5, 1.0, 2.0]
    params = [("alpha", [0.01, 0.1]), ("lumi", [0.1, 1.0, 10.0])]
    t = Sim5_FitsTable(fn, 5.0, 100.0, params, energies)
    gen = t.generator()
    for _ in range(2):                                        # two spectra, then the run is interrupted
        index, gi, gv = next(gen)
        t.write(index, 1.0 + index, [1, 2, 3], [3, 2, 1], flush=True)
    t2 = Sim5_FitsTable(fn, 5.0, 100.0, params, energies)     # a new run on the same file resumes after them
    todo = [i for (i, _, _) in t2.generator()]
    assert todo == [2, 3, 4, 5]
    assert t2.spectra["mdot"][:2].tolist() == [1.0, 2.0] and t2.spectra["Iv_f"][1].tolist() == [3.0, 2.0, 1.0]
    hdus = read_fits(fn)
    assert hdus[1]["header"]["EXTNAME"] == "META" and hdus[2]["rows"]["Iv_0"].shape == (6, 3)
    with pytest.raises(ValueError, match="metadata differ"):
        Sim5_FitsTable(fn, 5.0, 100.0, [("alpha", [0.01, 0.2]), ("lumi", [0.1, 1.0, 10.0])], energies)
    assert os.path.getsize(fn) % 2880 == 0
