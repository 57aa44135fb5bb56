"""Reader for the SDP directory format consumed by `sdpb -s <sdpDir>`: plain directory or
zip/tar archive, block data as JSON or Boost-binary (sdp_bin.py).

Mirrors the reference readers so the on-disk input format stays unchanged:
  control.json        -> num_blocks            (src/sdp_solve/Block_Info/read_block_info.cxx:38)
  block_info_<j>.json -> dim, num_points       (read_block_info.cxx:15-39)
  objectives.json     -> "constant", "b"       (src/sdp_solve/SDP/read_objectives.cxx:22-36)
  block_data_<j>.json -> c, B, bilinear_bases_even/odd
                                               (SDP/read_block_data/Json_Block_Data_Parser.hxx:26-36)
  normalization.json  -> "normalization"       (SDP/read_normalization.cxx:33-62, optional)

All numbers are kept as the decimal strings found on disk; conversion to the
multi-word device format happens behind the C ABI (include/sdpb_hip.h).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class SDPBlock:
    dim: int
    num_points: int
    bases_even: List[List[str]]   # rows x num_points
    bases_odd: List[List[str]]
    B: List[List[str]]            # P x N
    c: List[str]                  # P

    @property
    def schur_size(self) -> int:  # Block_Info.hxx:54-58
        return self.num_points * self.dim * (self.dim + 1) // 2

    @property
    def psd_sizes(self):          # Block_Info.hxx:86-96
        even = self.dim * ((self.num_points + 1) // 2)
        return even, self.dim * self.num_points - even

    @property
    def bases_heights(self):      # Block_Info.hxx:110-114
        d = self.num_points - 1
        return d // 2 + 1, (d + 1) // 2


@dataclass
class SDP:
    blocks: List[SDPBlock]
    b: List[str]
    constant: str
    normalization: Optional[List[str]] = None
    path: str = ""
    shape: Optional[tuple] = None  # (dims, num_points) when blocks are supplied lazily

    @property
    def J(self) -> int:
        return len(self.shape[0]) if self.shape else len(self.blocks)

    @property
    def N(self) -> int:
        return len(self.b)

    @property
    def dims(self):
        return list(self.shape[0]) if self.shape else [blk.dim for blk in self.blocks]

    @property
    def num_points(self):
        return list(self.shape[1]) if self.shape else [blk.num_points for blk in self.blocks]

    @property
    def P_total(self) -> int:
        return sum(K * m * (m + 1) // 2 for m, K in zip(self.dims, self.num_points))


def _flat(rows) -> str:
    """Row-major whitespace-joined text blob (the C ABI's *_txt arguments)."""
    if rows and isinstance(rows[0], list):
        return "\n".join(" ".join(r) for r in rows)
    return " ".join(rows)


def block_text(blk: SDPBlock):
    return (_flat(blk.bases_even).encode(), _flat(blk.bases_odd).encode(),
            _flat(blk.B).encode(), _flat(blk.c).encode())


class _Source:
    """An SDP directory or any archive of one (pmp2sdp --zip; the reference reads every libarchive
    format, src/sdpb_util/Archive_Reader.cxx; here zip, tar, tar.gz/bz2/xz)."""

    def __init__(self, path: str):
        import tarfile
        import zipfile
        self.path = path
        self.zip = self.tar = None
        if os.path.isdir(path):
            self.names = set(os.listdir(path))
        elif os.path.isfile(path) and zipfile.is_zipfile(path):
            self.zip = zipfile.ZipFile(path)
            self.names = {os.path.basename(n): n for n in self.zip.namelist() if not n.endswith("/")}
        elif os.path.isfile(path) and tarfile.is_tarfile(path):
            self.tar = tarfile.open(path)
            self.names = {os.path.basename(m.name): m for m in self.tar.getmembers() if m.isfile()}
        else:
            raise FileNotFoundError(f"SDP path does not exist or is neither a directory nor a zip/tar archive: {path}")

    def has(self, name: str) -> bool:
        return name in self.names

    def read(self, name: str) -> bytes:
        if name not in self.names:
            raise FileNotFoundError(f"{os.path.join(self.path, name)} not found")
        if self.zip:
            return self.zip.read(self.names[name])
        if self.tar:
            return self.tar.extractfile(self.names[name]).read()
        with open(os.path.join(self.path, name), "rb") as f:
            return f.read()


def read_sdp(path: str, precision: Optional[int] = None) -> SDP:
    """Read an SDP written by pmp2sdp: a directory or a zip/tar archive of it, block data in the
    JSON flavour (--outputFormat=json) or the Boost-binary one (the default, sdp_bin.py; needs
    `precision`, which must be the --precision the SDP was written with, read_block_data: SDP_Block_Data.cxx:41-43)."""
    src = _Source(path)
    num_blocks = int(json.loads(src.read("control.json"))["num_blocks"])
    obj = json.loads(src.read("objectives.json"))
    blocks = []
    for j in range(num_blocks):
        info = json.loads(src.read(f"block_info_{j}.json"))
        if src.has(f"block_data_{j}.json"):
            d = json.loads(src.read(f"block_data_{j}.json"))
        elif src.has(f"block_data_{j}.bin"):
            if precision is None:
                raise ValueError(f"{path}: block_data_{j}.bin needs the --precision the SDP was written with")
            from .sdp_bin import read_block_data_bin
            d = read_block_data_bin(src.read(f"block_data_{j}.bin"), precision)
        else:
            raise FileNotFoundError(f"{path}: neither block_data_{j}.json nor block_data_{j}.bin")
        blk = SDPBlock(dim=int(info["dim"]), num_points=int(info["num_points"]),
                       bases_even=d["bilinear_bases_even"], bases_odd=d["bilinear_bases_odd"],
                       B=d["B"], c=d["c"])
        he, ho = blk.bases_heights
        assert len(blk.bases_even) == he, (j, len(blk.bases_even), he)
        assert len(blk.bases_odd) == ho, (j, len(blk.bases_odd), ho)
        assert len(blk.c) == blk.schur_size and len(blk.B) == blk.schur_size
        blocks.append(blk)
    norm = None
    if src.has("normalization.json"):
        norm = json.loads(src.read("normalization.json")).get("normalization")
    return SDP(blocks=blocks, b=list(obj["b"]), constant=str(obj["constant"]),
               normalization=norm, path=path)


def write_sdp(sdp: SDP, path: str, command: str = "sdpb_amd synthetic generator", fmt: str = "json",
              precision: Optional[int] = None) -> None:
    """Write an SDP directory readable by the real sdpb: fmt "json" or "bin" (sdp_bin.py, needs precision)."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "control.json"), "w") as f:
        json.dump({"num_blocks": sdp.J, "command": command}, f, indent=2)
    with open(os.path.join(path, "objectives.json"), "w") as f:
        json.dump({"constant": sdp.constant, "b": sdp.b}, f, indent=2)
    for j, blk in enumerate(sdp.blocks):
        with open(os.path.join(path, f"block_info_{j}.json"), "w") as f:
            json.dump({"dim": blk.dim, "num_points": blk.num_points}, f, indent=2)
        d = {"bilinear_bases_even": blk.bases_even, "bilinear_bases_odd": blk.bases_odd, "c": blk.c, "B": blk.B}
        if fmt == "bin":
            from .sdp_bin import write_block_data_bin
            with open(os.path.join(path, f"block_data_{j}.bin"), "wb") as f:
                f.write(write_block_data_bin(d, precision))
        else:
            with open(os.path.join(path, f"block_data_{j}.json"), "w") as f:
                json.dump(d, f)
