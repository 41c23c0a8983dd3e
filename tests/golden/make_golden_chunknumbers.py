"""Golden vectors for ``calculate_chunk_numbers_for_box``: the reference's own function is lifted by AST from
/root/reference/syconn/extraction/object_extraction_wrapper.py (:23-55) and run on a stand-in chunk set (``chunk_size`` + ``coord_dict``:
all it reads) whose numbering is x-outermost like knossos_utils' ChunkDataset.  Only inputs and outputs are stored.

    python tests/golden/make_golden_chunknumbers.py      ->  tests/golden/g13_chunk_numbers.npz
"""
import ast
import itertools
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/syconn/extraction/object_extraction_wrapper.py'


def lift(path, name):
    ns = {'np': np}
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    return ns[name]


class CSet:
    def __init__(self, box, chunk):
        self.chunk_size = np.asarray(chunk)
        grid = [-(-b // c) for b, c in zip(box, chunk)]
        self.coord_dict = {}
        for n, pos in enumerate(itertools.product(*[range(g) for g in grid])):      # x outermost, z innermost
            self.coord_dict[tuple(int(p * c) for p, c in zip(pos, chunk))] = n


CASES = [  # box, chunk size, offset, size
    ((40, 30, 20), (10, 10, 10), (12, 0, 5), (10, 10, 10)),
    ((40, 30, 20), (10, 10, 10), (0, 0, 0), (40, 30, 20)),
    ((40, 30, 20), (10, 10, 10), (5, 5, 5), (1, 1, 1)),
    ((40, 30, 20), (10, 10, 10), (20, 10, 0), (20, 20, 20)),
    ((64, 48, 40), (16, 12, 8), (17, 13, 9), (30, 20, 25)),
    ((37, 29, 23), (16, 16, 16), (15, 0, 15), (2, 29, 2)),
]


def main():
    f = lift(REF, 'calculate_chunk_numbers_for_box')
    out = {'n_cases': np.array(len(CASES))}
    for i, (box, chunk, off, size) in enumerate(CASES):
        cs = CSet(box, chunk)
        lst, tr = f(cs, np.array(off), np.array(size))          # (the reference grows its arguments in place: fresh arrays)
        out.update({f'c{i}_box': np.array(box), f'c{i}_chunk': np.array(chunk), f'c{i}_offset': np.array(off), f'c{i}_size': np.array(size),
                    f'c{i}_list': np.array(lst, dtype=np.int64), f'c{i}_tr_keys': np.array(list(tr.keys()), dtype=np.int64),
                    f'c{i}_tr_vals': np.array(list(tr.values()), dtype=np.int64)})
        print(i, box, chunk, off, size, '->', lst)
    np.savez_compressed(os.path.join(HERE, 'g13_chunk_numbers.npz'), **out)


if __name__ == '__main__':
    main()
