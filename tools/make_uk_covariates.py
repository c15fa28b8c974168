#!/usr/bin/env python3
"""Derive the compact LAD covariate table shipped in covid19uk_amd/data/.

Run HERE only (needs /root/reference/data, which never travels to the GPU box):

    python tools/make_uk_covariates.py

Inputs (data, not source):  /root/reference/data/mergedflows.csv  (Flow,From,To)
                            /root/reference/data/c2019modagepop.csv (lad19cd, age bands)
Output: covid19uk_amd/data/uk_lad2019.npz
    lad19cd [380] <U9   sorted LAD codes
    C       [380,380] int32  commuting flows, C[dest, src]
    N       [380] int64      population (row sum over age bands)

The aggregation follows what the reference's loaders do with the same files:
duplicate (src,dest) rows are summed and the matrix is pivoted to [dest, src],
both axes sorted by LAD code (covid19uk/data/loaders.py:17-41); the population
is the sum over age columns sorted by code (loaders.py:44-58).
"""
import os
import sys

import numpy as np
import pandas as pd

REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                   "covid19uk_amd", "data", "uk_lad2019.npz")


def main():
    pop = pd.read_csv(os.path.join(REF, "c2019modagepop.csv"), index_col="lad19cd")
    pop = pop.sum(axis=1).sort_index()
    codes = np.array(pop.index, dtype="U9")
    idx = {c: i for i, c in enumerate(codes)}

    flows = pd.read_csv(os.path.join(REF, "mergedflows.csv"))
    flows = flows[flows["From"].isin(idx) & flows["To"].isin(idx)]
    C = np.zeros((len(codes), len(codes)), dtype=np.int64)
    src = flows["From"].map(idx).to_numpy()
    dst = flows["To"].map(idx).to_numpy()
    np.add.at(C, (dst, src), flows["Flow"].to_numpy())
    assert C.max() < 2**31
    np.savez_compressed(OUT, lad19cd=codes, C=C.astype(np.int32),
                        N=pop.to_numpy().astype(np.int64))
    print("wrote", os.path.normpath(OUT), C.shape, "density",
          float((C > 0).mean()), "bytes", os.path.getsize(OUT), file=sys.stderr)


if __name__ == "__main__":
    main()
