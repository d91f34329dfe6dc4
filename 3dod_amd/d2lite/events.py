"""detectron2.utils.events stand-in: a stack of scalar stores (no smoothing, no writers)."""
_STACK = []


class EventStorage:
    def __init__(self, start_iter=0):
        self.iter = start_iter
        self.scalars = {}
        self.history = {}

    def put_scalar(self, name, value, smoothing_hint=True):
        self.scalars[name] = value          # may be a 0-d device tensor: no host sync here
        h = self.history.setdefault(name, [])
        h.append((self.iter, value))
        if len(h) > 64:                      # a window, not the whole run (values may be device tensors)
            del h[:-32]

    def put_scalars(self, **kw):
        for k, v in kw.items():
            self.put_scalar(k, v)

    def put_image(self, name, img):
        pass

    def step(self):
        self.iter += 1

    def latest(self):
        return {k: float(v) for k, v in self.scalars.items()}

    def __enter__(self):
        _STACK.append(self)
        return self

    def __exit__(self, *a):
        assert _STACK[-1] is self
        _STACK.pop()


class JSONWriter:
    """detectron2.utils.events.JSONWriter [third-party, restated]: one json object per call with the latest value of every
    scalar in the storage, appended to `metrics.json`."""

    def __init__(self, json_file):
        import os
        os.makedirs(os.path.dirname(json_file) or ".", exist_ok=True)
        self._file = open(json_file, "a")

    def write(self, storage=None):
        import json
        storage = storage or get_event_storage()
        rec = {"iteration": storage.iter}
        for k, v in storage.scalars.items():
            try:
                rec[k] = float(v)
            except (TypeError, ValueError):
                pass
        self._file.write(json.dumps(rec, sort_keys=True) + "\n")
        self._file.flush()

    def close(self):
        self._file.close()


class CommonMetricPrinter:
    """one log line per call: iteration, total loss, the individual losses, learning rate"""

    def __init__(self, max_iter=None, logger=None):
        import logging
        self.max_iter, self.logger = max_iter, logger or logging.getLogger("cubercnn.events")

    def write(self, storage=None):
        storage = storage or get_event_storage()
        s = storage.latest()
        losses = "  ".join("{}: {:.4g}".format(k, v) for k, v in sorted(s.items()) if "loss" in k.lower() and k != "total_loss")
        self.logger.info(" iter: {}  total_loss: {:.4g}  {}  lr: {:.5g}".format(
            storage.iter, s.get("total_loss", float("nan")), losses, s.get("lr", float("nan"))))


_DEFAULT = EventStorage()


def get_event_storage():
    return _STACK[-1] if _STACK else _DEFAULT
