"""CPU oracle for the PNAPCSAFT forward + MAPE loss.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from this package.  The product package
(``gnn-epc-saft_amd/``) never imports it and has no CPU fallback.

PARITY UNPINNED: the reference (``/root/reference/gnnepcsaft/train/models.py``)
cannot be imported in this container (torch_geometric, ogb, lightning,
torchmetrics, ml_collections, feos, ... are absent), it ships no tests, no
golden vectors and no checkpoints.  The two restatements in this package
(``pna_torch`` = vectorised, PyG-equivalent op sequence; ``pna_loops`` =
independent per-node / per-edge float64 loops) are written from SURVEY.md
Appendix A and cross-check each other; nothing else pins them.
"""
