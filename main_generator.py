#!/usr/bin/env python3
"""Argument-compatible placeholder for the reference ``main_generator.py`` (flags of
``utils/args_parser_generator.py``).  The RAG generator (GNN/MLP fusion of retrieved demonstrations + greedy
decode, ``train/train_generator.py``, ``utils/Evaluation_generator.py``) CONSUMES this build's outputs
(``resources/retrieval_result/<ds>/*_index.gen``) but is the next scope row, not part of
the encode-and-retrieve hot path; it is not built yet and says so instead of silently doing something else."""
from rag4dyg_amd.cli_args import GENERATOR, parse


def main(argv=None):
    args = parse(GENERATOR, "main_generator.py", argv)
    raise NotImplementedError(
        f"main_generator ({args.dataset}): the RAG generator stage is not built in this round (SURVEY.md 8f-1). "
        "Its inputs -- retrieved top-K index files -- are produced by main_retriever.py / "
        "retrieval_data_annotation.py of this build in the reference's format.")


if __name__ == "__main__":
    main()
