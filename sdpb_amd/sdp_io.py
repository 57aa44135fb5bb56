"""Reader for the SDP directory format consumed by `sdpb -s <sdpDir>` (JSON flavour).

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


def read_sdp(path: str) -> SDP:
    """Read a JSON-format SDP directory written by pmp2sdp --outputFormat=json."""
    if not os.path.isdir(path):
        raise FileNotFoundError(f"SDP path does not exist or is not a directory: {path}")
    with open(os.path.join(path, "control.json")) as f:
        num_blocks = int(json.load(f)["num_blocks"])
    with open(os.path.join(path, "objectives.json")) as f:
        obj = json.load(f)
    blocks = []
    for j in range(num_blocks):
        with open(os.path.join(path, f"block_info_{j}.json")) as f:
            info = json.load(f)
        data_path = os.path.join(path, f"block_data_{j}.json")
        if not os.path.exists(data_path):
            raise FileNotFoundError(
                f"{data_path}: only the JSON block_data flavour is read here "
                "(Boost-binary .bin is listed as a follow-up in DESIGN.md)")
        with open(data_path) as f:
            d = json.load(f)
        blk = SDPBlock(dim=int(info["dim"]), num_points=int(info["num_points"]),
                       bases_even=d["bilinear_bases_even"], bases_odd=d["bilinear_bases_odd"],
                       B=d["B"], c=d["c"])
        he, ho = blk.bases_heights
        assert len(blk.bases_even) == he, (j, len(blk.bases_even), he)
        assert len(blk.bases_odd) == ho, (j, len(blk.bases_odd), ho)
        assert len(blk.c) == blk.schur_size and len(blk.B) == blk.schur_size
        blocks.append(blk)
    norm = None
    npath = os.path.join(path, "normalization.json")
    if os.path.exists(npath):
        with open(npath) as f:
            norm = json.load(f).get("normalization")
    return SDP(blocks=blocks, b=list(obj["b"]), constant=str(obj["constant"]),
               normalization=norm, path=path)


def write_sdp(sdp: SDP, path: str, command: str = "sdpb_amd synthetic generator") -> None:
    """Write an SDP in the same JSON directory format (readable by the real sdpb)."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "control.json"), "w") as f:
        json.dump({"num_blocks": sdp.J, "command": command}, f, indent=2)
    with open(os.path.join(path, "objectives.json"), "w") as f:
        json.dump({"constant": sdp.constant, "b": sdp.b}, f, indent=2)
    for j, blk in enumerate(sdp.blocks):
        with open(os.path.join(path, f"block_info_{j}.json"), "w") as f:
            json.dump({"dim": blk.dim, "num_points": blk.num_points}, f, indent=2)
        with open(os.path.join(path, f"block_data_{j}.json"), "w") as f:
            json.dump({"bilinear_bases_even": blk.bases_even,
                       "bilinear_bases_odd": blk.bases_odd, "c": blk.c, "B": blk.B}, f)
