#!/usr/bin/env python3
"""Drop-in for the reference ``get_train_query_time.py <dataset> <timestamp>``: writes ``resources/<dataset>_train_query_time.pt``
(the query times ``main_retriever.py --do_train`` reads).  Host-side; see ``rag4dyg_amd/query_time.py``."""
from rag4dyg_amd.query_time import main

if __name__ == "__main__":
    main()
