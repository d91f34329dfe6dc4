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


_DEFAULT = EventStorage()


def get_event_storage():
    return _STACK[-1] if _STACK else _DEFAULT
