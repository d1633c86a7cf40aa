"""Spectral model table in a FITS file -- the output format of the reference's Python layer
(ref: python/sim5fitstable.py:8-154, class Sim5_FitsTable), written and read back WITHOUT astropy.

Same interface and the same file layout: a primary HDU that carries the checksum of the parameter grids (keyword CRC,
ref :34-40, :69), a binary table META (NAME 16A, N 1J, GRID 1PE: reference mass and distance, the energy grid and one row
per parameter grid, the grids as variable-length float arrays on the heap, ref :72-92) and a binary table SPECTRA with one
row per point of the parameter space (mdot 1E, Iv_0 nE, Iv_f nE, ref :95-102).  `generator()` walks the parameter space
with the last grid changing fastest and skips rows that already hold a spectrum (mdot > 0), so an interrupted run
resumes (ref :109-141); `write()` stores a row, `save()` rewrites the file.

The reference hands the byte layout to astropy.io.fits, which is not in this image (nor is any other FITS library), so
this module writes FITS itself, to the standard (FITS 4.0: 2880-byte blocks, 80-character cards, big-endian data, 'P'
array descriptors + heap).  PARITY UNPINNED: no file written by the reference's own code path can be produced here to
compare with byte for byte; tests/test_fitstable.py holds the writer to the standard with an independent parser and to
its own reader.  Host-side I/O only -- nothing here computes rays.
"""
import hashlib
import os
import sys

import numpy as np

BLOCK = 2880


def _card(key, value=None, comment=""):
    if value is None:
        s = key
    elif isinstance(value, bool):
        s = "%-8s= %20s" % (key, "T" if value else "F")
    elif isinstance(value, (int, np.integer)):
        s = "%-8s= %20d" % (key, value)
    elif isinstance(value, float):
        s = "%-8s= %20s" % (key, ("%.15G" % value))
    else:
        v = "'%-8s'" % str(value).replace("'", "''")             # strings: at least 8 characters between the quotes
        s = "%-8s= %-20s" % (key, v)
    if comment:
        s += " / " + comment
    assert len(s) <= 80, s
    return s.ljust(80).encode("ascii")


def _header(cards):
    raw = b"".join(cards) + _card("END")
    return raw + b" " * (-len(raw) % BLOCK)


def _pad(data, fill=b"\0"):
    return data + fill * (-len(data) % BLOCK)


def params_crc(ref_mass, ref_dist, params, energies):
    """md5 over the textual form of the reference values and grids (ref :34-40; the reference's Python 2 str() of each
    value, here str() encoded as ASCII)"""
    m = hashlib.md5()
    m.update(str(ref_mass).encode())
    m.update(str(ref_dist).encode())
    for p in params:
        m.update((p[0] + str(p[1])).encode())
    for e in energies:
        m.update(str(e).encode())
    return m.hexdigest()


class Sim5_FitsTable:
    def __init__(self, filename, ref_mass, ref_dist, params, energies, create_if_missing=True):
        """
        Args (ref :10-22):
            ref_mass: reference BH mass [M_sun]
            ref_dist: reference BH distance [pc]
            params:   parameter grids [(grid_name, grid_values), ...]; the last one changes fastest
            energies: energy grid [keV]
        """
        self.filename = filename
        self.params = [(str(n), np.asarray(v, dtype=np.float64)) for (n, v) in params]
        self.energies = np.asarray(energies, dtype=np.float64)
        self.ref_mass, self.ref_dist = float(ref_mass), float(ref_dist)
        self.crc = params_crc(ref_mass, ref_dist, params, energies)
        self.total_grid_size = 1
        for _, v in self.params:
            self.total_grid_size *= len(v)
        ne = len(self.energies)
        self.spectra = np.zeros(self.total_grid_size, dtype=[("mdot", ">f4"), ("Iv_0", ">f4", (ne,)), ("Iv_f", ">f4", (ne,))])
        if os.path.isfile(filename):
            sys.stderr.write("Sim5_FitsTable: opening existing fits file %s\n" % filename)
            hdus = read_fits(filename)
            if hdus[0]["header"].get("CRC") != self.crc:
                raise ValueError("Sim5_FitsTable: cannot open %s, metadata differ" % filename)
            spec = [h for h in hdus if h["header"].get("EXTNAME") == "SPECTRA"]
            if not spec or len(spec[0]["rows"]) != self.total_grid_size:
                raise ValueError("Sim5_FitsTable: %s has no SPECTRA table of %d rows" % (filename, self.total_grid_size))
            self.spectra[:] = spec[0]["rows"]
        elif create_if_missing:
            sys.stderr.write("Sim5_FitsTable: creating new fits file\n")
            self.save()
        sys.stderr.write("Sim5_FitsTable: total grid size = %d\n" % self.total_grid_size)

    def generator(self):
        """(index, grid indexes, grid values) of every row without a spectrum, the last grid changing fastest (ref :109-141)"""
        index = 0
        while index < self.total_grid_size:
            while index < self.total_grid_size and self.spectra["mdot"][index] > 0.0:
                index += 1
            if index >= self.total_grid_size:
                break
            gindices = np.zeros(len(self.params), dtype=int)
            gvalues = np.zeros(len(self.params))
            n0 = self.total_grid_size
            for i, (_, vals) in enumerate(self.params):
                n0 //= len(vals)
                gindices[i] = index // n0 % len(vals)
                gvalues[i] = vals[gindices[i]]
            yield (index, gindices, gvalues)
            index += 1

    def write(self, index, mdot, Iv_0, Iv_f, flush=False):
        """spectrum of the grid point `index` (ref :144-152)"""
        self.spectra["mdot"][index] = mdot
        self.spectra["Iv_0"][index] = np.asarray(Iv_0, dtype=np.float64)
        self.spectra["Iv_f"][index] = np.asarray(Iv_f, dtype=np.float64)
        if flush:
            self.save()

    # ---- the file ----
    def _meta_hdu(self):
        rows = [("REF_MASS", np.array([self.ref_mass])), ("REF_DIST", np.array([self.ref_dist])), ("ENERGIES", self.energies)]
        rows += [(n.upper(), v) for (n, v) in self.params]
        heap = b""
        table = b""
        longest = 0
        for name, vals in rows:
            arr = np.asarray(vals, dtype=">f4")
            table += name.encode("ascii")[:16].ljust(16) + np.array([len(arr)], ">i4").tobytes()
            table += np.array([len(arr), len(heap)], ">i4").tobytes()           # 'P' descriptor: element count, heap offset
            heap += arr.tobytes()
            longest = max(longest, len(arr))
        cards = [_card("XTENSION", "BINTABLE", "binary table extension"), _card("BITPIX", 8), _card("NAXIS", 2),
                 _card("NAXIS1", 28, "bytes per row"), _card("NAXIS2", len(rows), "rows"), _card("PCOUNT", len(heap), "size of the heap"),
                 _card("GCOUNT", 1), _card("TFIELDS", 3),
                 _card("TTYPE1", "NAME"), _card("TFORM1", "16A"), _card("TTYPE2", "N"), _card("TFORM2", "1J"),
                 _card("TTYPE3", "GRID"), _card("TFORM3", "1PE(%d)" % longest), _card("EXTNAME", "META")]
        return _header(cards) + _pad(table + heap)

    def _spectra_hdu(self):
        ne = len(self.energies)
        cards = [_card("XTENSION", "BINTABLE", "binary table extension"), _card("BITPIX", 8), _card("NAXIS", 2),
                 _card("NAXIS1", 4 + 8 * ne, "bytes per row"), _card("NAXIS2", self.total_grid_size, "rows"), _card("PCOUNT", 0),
                 _card("GCOUNT", 1), _card("TFIELDS", 3),
                 _card("TTYPE1", "mdot"), _card("TFORM1", "1E"), _card("TTYPE2", "Iv_0"), _card("TFORM2", "%dE" % ne),
                 _card("TTYPE3", "Iv_f"), _card("TFORM3", "%dE" % ne), _card("EXTNAME", "SPECTRA")]
        return _header(cards) + _pad(self.spectra.tobytes())

    def save(self):
        """rewrite the file: primary HDU with the CRC, META, SPECTRA (ref :155-162)"""
        primary = _header([_card("SIMPLE", True, "conforms to FITS standard"), _card("BITPIX", 8), _card("NAXIS", 0),
                           _card("EXTEND", True), _card("CRC", self.crc)])
        tmp = self.filename + ".tmp"
        with open(tmp, "wb") as fh:
            fh.write(primary + self._meta_hdu() + self._spectra_hdu())
        os.replace(tmp, self.filename)


def _parse_header(raw, pos):
    hdr = {}
    while True:
        block = raw[pos:pos + BLOCK]
        if len(block) < BLOCK:
            raise ValueError("truncated FITS header")
        pos += BLOCK
        done = False
        for i in range(0, BLOCK, 80):
            card = block[i:i + 80].decode("ascii")
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if card[8:10] != "= ":
                continue
            val = card[10:].split(" / ")[0].strip()
            if val.startswith("'"):
                hdr[key] = val[1:val.rindex("'")].replace("''", "'").rstrip()
            elif val in ("T", "F"):
                hdr[key] = val == "T"
            else:
                hdr[key] = float(val) if any(c in val for c in ".EeD") else int(val)
        if done:
            return hdr, pos


def read_fits(filename):
    """[{header, rows, heap}] for the primary HDU and the binary tables of a file written by this module (or any FITS file
    made of a data-less primary HDU and binary tables whose fixed-width columns are A, J, E and P descriptors)"""
    raw = open(filename, "rb").read()
    pos = 0
    out = []
    while pos < len(raw):
        hdr, pos = _parse_header(raw, pos)
        hdu = {"header": hdr, "rows": None, "heap": b""}
        if hdr.get("XTENSION") == "BINTABLE":
            n1, n2, pc = hdr["NAXIS1"], hdr["NAXIS2"], hdr.get("PCOUNT", 0)
            fields = []
            for i in range(1, hdr["TFIELDS"] + 1):
                tf = hdr["TFORM%d" % i].split("(")[0]
                rep = int("".join(c for c in tf[:-1] if c.isdigit()) or "1") if not tf.endswith(("PE", "PJ")) else 1
                code = tf[-2:] if tf[-2:] in ("PE", "PJ") else tf[-1]
                name = hdr["TTYPE%d" % i]
                if code == "A":
                    fields.append((name, "S%d" % rep))
                elif code == "J":
                    fields.append((name, ">i4") if rep == 1 else (name, ">i4", (rep,)))
                elif code == "E":
                    fields.append((name, ">f4") if rep == 1 else (name, ">f4", (rep,)))
                elif code in ("PE", "PJ"):
                    fields.append((name, ">i4", (2,)))
                else:
                    raise ValueError("column format %r" % hdr["TFORM%d" % i])
            dt = np.dtype(fields)
            assert dt.itemsize == n1, (dt.itemsize, n1)
            hdu["rows"] = np.frombuffer(raw[pos:pos + n1 * n2], dtype=dt).copy()
            hdu["heap"] = raw[pos + n1 * n2:pos + n1 * n2 + pc]
            size = n1 * n2 + pc
            pos += size + (-size % BLOCK)
        out.append(hdu)
    return out
