"""numpy arrays in an sqlite3 table, API of plancklens/helpers/sql.py (`npdb` :28-66): the spectra cache of qecl."""
import io
import os
import sqlite3

import numpy as np

from . import mpi


def _adapt(arr):
    out = io.BytesIO()
    np.save(out, arr)
    return memoryview(out.getvalue())


def _convert(blob):
    return np.load(io.BytesIO(blob))


sqlite3.register_adapter(np.ndarray, _adapt)
sqlite3.register_converter("ARRAY", _convert)


class npdb(object):
    def __init__(self, fname, idtype="STRING"):
        if not os.path.exists(fname) and mpi.rank == 0:
            con = sqlite3.connect(fname, detect_types=sqlite3.PARSE_DECLTYPES, timeout=3600)
            con.execute("CREATE TABLE npdb (id %s PRIMARY KEY, arr ARRAY)" % idtype)
            con.commit()
            con.close()
        mpi.barrier()
        self.con = sqlite3.connect(fname, timeout=3600., detect_types=sqlite3.PARSE_DECLTYPES)

    def add(self, idx, vec):
        try:
            assert self.get(idx) is None
            self.con.execute("INSERT INTO npdb (id,  arr) VALUES (?,?)", (idx, np.asarray(vec).reshape((1, len(vec)))))
            self.con.commit()
        except Exception:
            print("npdb add failed!")

    def remove(self, idx):
        try:
            assert self.get(idx) is not None
            self.con.execute("DELETE FROM npdb WHERE id=?", (idx,))
            self.con.commit()
        except Exception:
            print("npdb remove failed!")

    def get(self, idx):
        cur = self.con.cursor()
        cur.execute("SELECT arr FROM npdb WHERE id=?", (idx,))
        data = cur.fetchone()
        cur.close()
        return None if data is None else data[0].flatten()
