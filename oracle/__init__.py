"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the HOP generator hot path.

`oracle/` restates, in plain functional PyTorch on the CPU, the arithmetic of the
reference's hot path (model/HOP.py, model/gwnet.py, train_eval/train_llm.py,
model/multimodal_context_net.py::ConvDiscriminator) so that the HIP product path
can be checked against it.  It is pinned to the reference by the golden vectors in
`tests/golden/` (made by `tools/make_golden.py`, which imports the real reference
in the build container; see DESIGN.md "Oracle").

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package.  The product package never does: its ops raise when the HIP
library is missing instead of falling back to anything here.
"""
