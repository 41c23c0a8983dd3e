"""Worker entry of the dense prediction (process boundary of /root/reference/syconn/batchjob_scripts/
batchjob_predict_dense.py:1-20): ``python batchjob_predict_dense.py <in.pkl> <out.pkl>`` -- `in.pkl` is a stream of
pickles, one per element of the argument tuple; `out.pkl` (``pickle.dump(None)``) is the done-marker."""
import pickle as pkl
import sys

from syconn_amd.handler.prediction import dense_predictor

path_storage_file = sys.argv[1]
path_out_file = sys.argv[2]

with open(path_storage_file, 'rb') as f:
    args = []
    while True:
        try:
            args.append(pkl.load(f))
        except EOFError:
            break

dense_predictor(args)

with open(path_out_file, "wb") as f:
    pkl.dump(None, f)
