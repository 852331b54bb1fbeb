"""The many-sequence (16-bit) decode step alone, for `rocprofv3 --pmc` passes: one greedy generate of `rows` sequences with the
decode steps launched eagerly (graph=0) or as hipGraph replays (graph=1).  With graph=0 the step's three row ranges are launched
one after the other on ONE stream (RGRG_DECODE_CHAINS=-3 unless the variable is already set): the same kernels on the same row
counts as the product's step, without the concurrent streams that the counter mode does not survive either.  tools/collect_profiles.sh profiles it instead of the
whole bench.py command line, whose counter pass dies inside rocprofv3 on this image (profiles/r04_pmc_fetch_b32_rocprofv3_crash.log);
run with graph=1 under --pmc it is the short reproducer of that crash, with graph=0 it gives the FETCH_SIZE / WRITE_SIZE of the
real decode process (tools/pmc_traffic.py: keys gemm_S<rows>_bf16 / attn_S<rows>_bf16).
Usage: python tools/decode_pmc_probe.py [rows=923] [tokens=128] [graph=0]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rgrg_amd  # noqa: E402
from rgrg_amd import synth  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 923
    tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    graph = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    if not graph:
        os.environ.setdefault("RGRG_DECODE_CHAINS", "-3")   # read when the decoder is created
    m = rgrg_amd.ReportGenerationModel(pretrain_without_lm_model=True)
    m.load_state_dict(synth.make_state_dict(0, "bench"))
    m.to("cuda:0").eval()
    g = torch.Generator().manual_seed(99)
    feats = torch.randn((rows, 1024), generator=g).to("cuda:0")
    eng = m.engine()
    ids = eng.greedy_decode(feats, tokens, use_graph=graph, bf16=1)
    torch.cuda.synchronize()
    print("generated", tuple(ids.shape), "graph", graph, flush=True)
    eng.close()


if __name__ == "__main__":
    main()
