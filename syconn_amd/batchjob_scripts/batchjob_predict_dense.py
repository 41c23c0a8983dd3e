"""Worker entry of the dense prediction: ``python batchjob_predict_dense.py <in.pkl> <out.pkl>``.

Process-boundary contract of the reference's script of the same name (/root/reference/syconn/batchjob_scripts/
batchjob_predict_dense.py:1-20, written by /root/reference/syconn/mp/batchjob_utils.py:227-232): ``<in.pkl>`` holds the
elements of ``dense_predictor``'s argument tuple as CONCATENATED pickles (one ``pickle.dump`` per element, no enclosing
container); the worker writes ``<out.pkl>`` = ``pickle.dump(None)`` when it is done -- its existence is what the dispatcher
checks (a missing file raises ``ValueError`` there)."""
import pickle
import sys


def read_pickle_stream(path):
    """All objects of a file of back-to-back pickles, in order."""
    items = []
    with open(path, 'rb') as stream:
        unpickler_eof = False
        while not unpickler_eof:
            try:
                items.append(pickle.load(stream))
            except EOFError:
                unpickler_eof = True
    return items


def main(argv):
    if len(argv) != 3:
        raise SystemExit('usage: batchjob_predict_dense.py <in.pkl> <out.pkl>')
    job_file, done_file = argv[1], argv[2]
    from syconn_amd.handler.prediction import dense_predictor
    dense_predictor(read_pickle_stream(job_file))
    with open(done_file, 'wb') as marker:
        pickle.dump(None, marker)


if __name__ == '__main__':
    main(sys.argv)
