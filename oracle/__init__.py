"""CPU oracle for the RGRG inference hot path.  TEST INFRASTRUCTURE ONLY.

This package is a torch-CPU fp32 restatement of the reference's
``ReportGenerationModel.generate`` path (ttanida/rgrg,
``src/full_model/report_generation_model.py:212-276``).  It is imported only
by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg, as the checker - never by ``rgrg_amd`` (the product), which must fail
loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * everything the reference itself authors (top-1-per-class post-processing,
    region selection, pseudo-self-attention GPT-2, greedy loop, generate()
    orchestration) is PINNED: ``tests/golden/make_golden.py`` imports the real
    reference modules in the build container, loads the same seeded synthetic
    state dict into them and dumps their outputs as fixtures which
    ``tests/test_oracle_golden.py`` compares this restatement against.
  * the arithmetic the reference delegates to torchvision==0.13.1 (ResNet-50,
    AnchorGenerator, RPNHead, BoxCoder, filter_proposals/nms, roi_align,
    TwoMLPHead, FastRCNNPredictor) is restated in ``oracle/tv013.py`` from the
    documented 0.13.1 semantics.  torchvision is not installed in the build
    image and the reference holds no tests/golden vectors, so that part is
    "PARITY UNPINNED" beyond hand-computable known-answer tests.
"""
