"""Signature of the device code inside ``libwcn_hip.so``: (mangled name, code size) of every ``wcn::`` function of the embedded
gfx950 code objects, hashed.  `bench.py` stamps the committed PMC summaries with it (`tools/rocpd_stats.py traffic`) and reports
the static ``roofline.traffic`` figure only while the running library still has the kernels the counters were collected on; a
kernel edit that is not followed by `tools/collect_profiles.sh` nulls the field instead of shipping a stale number.  Names and
sizes - not bytes: the fat binary embeds build paths, the same sources built in two directories differ in bytes and agree here."""
import hashlib
import struct
from typing import Dict, Iterable, List, Optional, Tuple

# kernels of the headline step's phases (substring of the mangled name -> phase)
PHASE_KERNELS = {
    "kmap": ("cell_prepare", "cell_insert", "cell_finish", "cell_neighbors", "kmap_tally", "kmap_scan", "kmap_scatter", "rs_kernel",
             "rs_pairs", "repair_row"),
    "fwd": ("gather_gemm_cs_kernel",),
    "dgrad": ("gather_gemm_cs_kernel",),
    "wgrad": ("wgrad_mfma_kernel", "wgrad_reduce"),
}


def _sections(b: bytes, base: int = 0):
    shoff = struct.unpack_from("<Q", b, base + 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, base + 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, _addr, offset, size, link, _info, _align, entsize = struct.unpack_from(
            "<IIQQQQIIQQ", b, base + shoff + i * shentsize)
        secs.append(dict(name=name, type=typ, offset=offset, size=size, link=link, entsize=entsize))
    return secs, shstrndx


def _cstr(b: bytes, off: int) -> str:
    return b[off : b.index(b"\0", off)].decode(errors="replace")


def device_functions(path: str) -> List[Tuple[str, int]]:
    """Sorted ``(mangled name, code bytes)`` of the ``wcn::`` functions in the gfx950 code objects of ``path``."""
    with open(path, "rb") as f:
        b = f.read()
    if b[:4] != b"\x7fELF" or b[4] != 2:
        return []
    secs, shstr = _sections(b)
    stroff = secs[shstr]["offset"]
    out = set()
    for s in secs:
        if _cstr(b, stroff + s["name"]) != ".hip_fatbin":
            continue
        lo, hi = s["offset"], s["offset"] + s["size"]
        pos = lo
        while True:
            pos = b.find(b"\x7fELF\x02\x01\x01", pos, hi)
            if pos < 0:
                break
            try:
                if struct.unpack_from("<H", b, pos + 0x12)[0] == 224:  # EM_AMDGPU
                    dsecs, _ = _sections(b, pos)
                    for d in dsecs:
                        if d["type"] != 2 or d["entsize"] != 24:  # SHT_SYMTAB
                            continue
                        strtab = dsecs[d["link"]]
                        for i in range(d["size"] // 24):
                            nm, info, _other, shndx, _value, size = struct.unpack_from("<IBBHQQ", b, pos + d["offset"] + 24 * i)
                            if (info & 0xF) == 2 and shndx != 0:  # STT_FUNC, defined
                                name = _cstr(b, pos + strtab["offset"] + nm)
                                if name.startswith("_ZN3wcn"):
                                    out.add((name, int(size)))
            except (struct.error, ValueError, IndexError):
                pass
            pos += 4
    return sorted(out)


def signature(functions: Iterable[Tuple[str, int]], needles: Optional[Iterable[str]] = None) -> str:
    sel = [f for f in functions if needles is None or any(n in f[0] for n in needles)]
    return hashlib.sha256(repr(sel).encode()).hexdigest()[:16]


def phase_signatures(path: str) -> Dict[str, str]:
    """``{phase: signature}`` for the phases of the headline step, plus ``"all"``."""
    fn = device_functions(path)
    out = {k: signature(fn, v) for k, v in PHASE_KERNELS.items()}
    out["all"] = signature(fn)
    return out
