"""A dict-like result store for multi-process tests, backed by files: `store[rank] = obj` in a worker,
`store[rank]` in the parent.  Replaces `mp.Manager().dict()` in the GPU tests: a Manager is a server process
FORKED from the pytest process, and forking a process that has initialised the HIP runtime is not safe (the
manager crashed with a segmentation fault while pickling on the GPU box)."""
import os
import pickle
import tempfile


class ResultStore:
    def __init__(self):
        self.path = tempfile.mkdtemp(prefix="tbe_results_")

    def _file(self, key):
        return os.path.join(self.path, f"{key}.pkl")

    def __setitem__(self, key, value):
        tmp = self._file(key) + ".tmp"
        with open(tmp, "wb") as fh:
            pickle.dump(value, fh, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(tmp, self._file(key))

    def __getitem__(self, key):
        with open(self._file(key), "rb") as fh:  # written by this test's own worker
            return pickle.load(fh)

    def __contains__(self, key):
        return os.path.exists(self._file(key))
