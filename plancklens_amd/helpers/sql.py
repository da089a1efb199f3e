"""numpy arrays and floats in sqlite3 tables, API of plancklens/helpers/sql.py (`npdb` :28-66, `fldb` :68-106): the spectra cache of qecl."""
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


class _table(object):
    """One (id, value) sqlite3 table in a file of its own: rank 0 creates it, every rank opens it.  Failed adds / removes are
    reported, not raised, as in the reference (sql.py:42-56,82-96)."""
    name, column, sqltype = None, None, None

    def __init__(self, fname, idtype="STRING"):
        if not os.path.exists(fname) and mpi.rank == 0:
            con = sqlite3.connect(fname, detect_types=sqlite3.PARSE_DECLTYPES, timeout=3600)
            con.execute("CREATE TABLE %s (id %s PRIMARY KEY, %s %s)" % (self.name, idtype, self.column, self.sqltype))
            con.commit()
            con.close()
        mpi.barrier()
        self.con = sqlite3.connect(fname, timeout=3600., detect_types=sqlite3.PARSE_DECLTYPES)

    def _pack(self, value):
        return value

    def _unpack(self, value):
        return value

    def add(self, idx, value):
        try:
            assert self.get(idx) is None
            self.con.execute("INSERT INTO %s (id,  %s) VALUES (?,?)" % (self.name, self.column), (idx, self._pack(value)))
            self.con.commit()
        except Exception:
            print("%s add failed!" % self.name)

    def remove(self, idx):
        try:
            assert self.get(idx) is not None
            self.con.execute("DELETE FROM %s WHERE id=?" % self.name, (idx,))
            self.con.commit()
        except Exception:
            print("%s remove failed!" % self.name)

    def get(self, idx):
        cur = self.con.cursor()
        cur.execute("SELECT %s FROM %s WHERE id=?" % (self.column, self.name), (idx,))
        data = cur.fetchone()
        cur.close()
        return None if data is None else self._unpack(data[0])


class npdb(_table):
    """numpy arrays by id (sql.py:28-66): table `npdb`, column `arr` of the registered ARRAY type, stored as a (1, n) array and
    handed back flattened -- files are interchangeable with the reference's"""
    name, column, sqltype = 'npdb', 'arr', 'ARRAY'

    def _pack(self, vec):
        return np.asarray(vec).reshape((1, len(vec)))

    def _unpack(self, arr):
        return arr.flatten()


class fldb(_table):
    """floats by id (sql.py:68-106; used by the n1 library's cache): table `fldb`, column `fl` REAL"""
    name, column, sqltype = 'fldb', 'fl', 'REAL'
