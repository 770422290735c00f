"""CPU oracle for the RAG4DyG encode-and-retrieve hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``rag4dyg_amd``) never imports this package and fails loudly
when the HIP extension is missing.

Every function here is our own CPU restatement (torch-CPU fp32 for the
floating-point encoder / scoring, python sets + numpy for the integer Jaccard
pass) of the reference algorithm, citing the reference ``file:line`` it
follows.  Parity status: PINNED -- ``oracle/gen_golden.py`` imports the
reference's own modules from ``/root/reference`` in the build container, runs
them on CPU, and commits the resulting input/output vectors under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every oracle
function against those vectors (the reference itself ships no tests, golden
vectors or fixtures -- SURVEY.md section 4).
"""
