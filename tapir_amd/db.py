"""Mirror of tapir/db.py: the phylogenetic-informativeness.sqlite writer.

The DDL strings are the reference's own (tapir/db.py:19-30), whitespace included, because sqlite stores
them verbatim in sqlite_master.sql and the drop-in contract is a byte-identical schema.  Differences:
rows go in with executemany inside one transaction (the reference issues one execute per row, :51-60), and
an existing database is replaced without the interactive prompt (the reference's prompt always answers
yes: `answer == "Y" or "YES"`, :34)."""
import os
import sqlite3

DDL = [
    "CREATE TABLE loci (id INTEGER PRIMARY KEY AUTOINCREMENT, locus TEXT)",
    '''CREATE TABLE net (id INT, time INT, pi FLOAT,
            FOREIGN KEY(id) REFERENCES loci(id) DEFERRABLE INITIALLY
            DEFERRED)''',
    '''CREATE TABLE discrete (id INT, time INT, pi FLOAT, 
            FOREIGN KEY(id) REFERENCES loci(id) DEFERRABLE INITIALLY
            DEFERRED)''',
    '''CREATE TABLE interval (id INT, interval TEXT, pi FLOAT,
            error FLOAT, FOREIGN KEY(id) REFERENCES loci(id) DEFERRABLE
            INITIALLY DEFERRED)''',
]


def create_probe_db(db_name):
    """Create the four tables; returns (conn, cursor) like tapir/db.py:15-42."""
    if os.path.exists(db_name):
        os.remove(db_name)
    conn = sqlite3.connect(db_name)
    c = conn.cursor()
    c.execute("PRAGMA foreign_keys = ON")
    for stmt in DDL:
        c.execute(stmt)
    return conn, c


def locus_name(path):
    """Name stored in loci.locus: basename minus its last extension (tapir/db.py:47-48)."""
    return os.path.splitext(os.path.basename(path))[0]


def insert_pi_data(conn, c, pis):
    """pis: iterable of worker()-style tuples (name, rates, mean_rate, pi, pi_net, times, epochs); only
    name, pi_net, times and epochs are stored (tapir/db.py:44-61)."""
    for locus in pis:
        name, rates, mean_rate, pi, pi_net, times, epochs = locus
        c.execute("INSERT INTO loci(locus) VALUES (?)", (locus_name(name),))
        key = c.lastrowid
        c.executemany("INSERT INTO net VALUES (?,?,?)", [(key, k, float(v)) for k, v in enumerate(pi_net)])
        if times:
            c.executemany("INSERT INTO discrete VALUES (?,?,?)", [(key, int(k), float(v)) for k, v in times.items()])
        if epochs:
            c.executemany("INSERT INTO interval VALUES (?,?,?,?)",
                          [(key, k, float(v['sum(integral)']), float(v['sum(error)'])) for k, v in epochs.items()])
    return


def insert_tables(conn, c, names, tables, T, times, intervals):
    """Bulk form for the batch engine: tables[l] = [net(T) | disc(n_t) | integral(n_i) | error(n_i)].  One executemany per
    table over columns prepared with numpy (C4: 5.4 M rows; a list comprehension per locus was half of the stage's time).
    Row order inside each table is locus by locus, as the reference's loop writes them (tapir/db.py:44-61)."""
    import numpy as np
    L = len(names)
    if L == 0:
        return
    n_t, n_i = len(times), len(intervals)
    labels = ["{0}-{1}".format(a, b) for a, b in intervals]
    tables = np.asarray(tables, dtype=np.float64)
    c.executemany("INSERT INTO loci(locus) VALUES (?)", [(locus_name(n),) for n in names])
    keys = [r[0] for r in c.execute("SELECT id FROM loci ORDER BY id DESC LIMIT ?", (L,)).fetchall()][::-1]
    keys = np.asarray(keys, dtype=np.int64)
    if T:
        c.executemany("INSERT INTO net VALUES (?,?,?)",
                      zip(np.repeat(keys, T).tolist(), np.tile(np.arange(T), L).tolist(), tables[:, :T].reshape(-1).tolist()))
    if n_t:
        c.executemany("INSERT INTO discrete VALUES (?,?,?)",
                      zip(np.repeat(keys, n_t).tolist(), np.tile(np.asarray(times, dtype=np.int64), L).tolist(),
                          tables[:, T:T + n_t].reshape(-1).tolist()))
    if n_i:
        c.executemany("INSERT INTO interval VALUES (?,?,?,?)",
                      zip(np.repeat(keys, n_i).tolist(), labels * L, tables[:, T + n_t:T + n_t + n_i].reshape(-1).tolist(),
                          tables[:, T + n_t + n_i:T + n_t + 2 * n_i].reshape(-1).tolist()))
