"""Host-side bounds of the SASRec step's workspaces (no GPU): every region the library carves out of the caller's backward workspace
(csrc/enc_tile_prep.h: enc_bwd_ws, the ONE place that does it) must hold the largest index a kernel can form into it, and the last region
must end inside what the size query returns -- for the shapes that stress each region: one sequence, the tile cap of the dK / dV inboxes
(5 120 tiles), the looped form of the tile kernels (> 1 024 tiles), D = 128, the most blocks, the finest split of the weight-gradient
contractions (40 from 1 024 sequences on), misaligned buffers.

The expected sizes below are restated from the kernels' index expressions, not from the carving code:
  slabs            enc_tile_step_k writes slab[tile][l][12][D] (one per TILE in the tile form), enc_step_k slab[workgroup <= 1024][..]
  matrix partials  wg_matrix_job: part[((l * 6 + m) * nsplit + split) * D * D + ...], nsplit <= 40
  position partials wg_pos_job:   ppart[(p * 144 + group) * D + col], p < S <= 64
  gradient tape    tile kernels:  gtape[(l * 6 + m) * NR * D + row * D + col], NR = 16 * max_tiles
  weight fragments tl_prep_thread: L x 6 matrices x 2 orientations x [ns][ns / 2][2 planes][64 lanes] x 16 B, then (10 L + 2) x D floats
  inboxes          tile kernels:  xch[(tile, l)] x 3 x 2 x ns x 256 floats for min(max_tiles, 5 120) tiles
"""
import ctypes
import itertools

import pytest

from recboard_amd import lib


def _layout(L, B, S, D, nl, base):
    out = (ctypes.c_uint64 * 7)()
    assert L.re_sasrec_encoder_bwd_workspace_layout(B, S, D, nl, base, out) == 0
    return list(out)


CASES = [(1, 1), (1, 50), (7, 17), (512, 50), (513, 50), (1024, 50), (1280, 64), (2048, 50), (4096, 50), (8192, 50), (20000, 33), (3, 64)]


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("nl", [1, 2, 4])
def test_backward_workspace_regions_hold_what_the_kernels_index(D, nl):
    L = lib.load()
    ns = D // 16
    for (B, S), base in itertools.product(CASES, (1 << 20, (1 << 20) + 4, (1 << 20) + 252)):
        total = L.re_sasrec_encoder_bwd_workspace_bytes(B, S, D, nl)
        slab, wpart, ppart, gtape, wf, xch, end = _layout(L, B, S, D, nl, base)
        mt = B * ((S + 15) // 16)
        assert slab == 0 and slab <= wpart <= ppart <= gtape <= wf <= xch <= end, (B, S, base)
        assert end <= total, (B, S, D, nl, base, end, total)
        assert (base + wf) % 256 == 0 and (base + xch) % 256 == 0
        assert wpart - slab >= 4 * max(mt, 1024) * nl * 12 * D
        nsplit = 40 if (D == 64 and B >= 1024) else (12 if D == 128 else 24)
        assert ppart - wpart >= 4 * nl * 6 * nsplit * D * D
        assert gtape - ppart >= 4 * S * 144 * D
        assert wf - gtape >= 4 * nl * 6 * 16 * mt * D
        frag_words = ns * (ns // 2) * 2 * 64 * 4
        assert xch - wf >= 4 * nl * 6 * 2 * frag_words + 4 * (10 * nl + 2) * D
        assert end - xch >= 4 * min(mt, 5120) * nl * 3 * 2 * ns * 256


def test_layout_query_rejects_what_the_kernels_do_not_run():
    L = lib.load()
    out = (ctypes.c_uint64 * 7)()
    for B, S, D, nl in ((0, 50, 64, 2), (4, 0, 64, 2), (4, 65, 64, 2), (4, 50, 96, 2), (4, 50, 64, 0), (4, 50, 64, 5)):
        assert L.re_sasrec_encoder_bwd_workspace_layout(B, S, D, nl, 1 << 20, out) != 0
    assert L.re_sasrec_encoder_bwd_workspace_layout(4, 50, 64, 2, 1 << 20, None) != 0


def test_plan_buffer_has_room_for_spans_placements_and_the_hand_over_words():
    """re_sasrec_plan_bytes: header + work items + row map (2 x 16 x max_tiles words) + a span and a placement word per sequence + 64 spare words
    (csrc/enc_plan_body.h: the span hand-over's flag, arrival count and token count are the first three of them)."""
    L = lib.load()
    for B, S in CASES:
        mt = B * ((S + 15) // 16)
        got = L.re_sasrec_plan_bytes(B, S)
        assert got >= 4 * (mt + 2 * 16 * mt + 2 * B + 64)
        assert got % 4 == 0
