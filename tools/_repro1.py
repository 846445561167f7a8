import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import importlib
m = importlib.import_module('test_gpu_filters_multidevice')
import duckdb_arrow_amd as da
con = da.Connection(0)
t = m._table()
path = '/tmp/t.arrows'
m._write(t, path)
for compact in (False, True):
    for expr in m.EXPRS[:3]:
        rel = con.read_arrow(path, filter_compact=compact).project(["k", "i32", "s", "dec"]).filter(expr)
        try:
            cols = rel.fetch_columns()
            print(compact, expr, len(cols[0]), int(m._numpy_eval(expr, t).sum()))
        except Exception as e:
            print('ERR', compact, expr, e)
