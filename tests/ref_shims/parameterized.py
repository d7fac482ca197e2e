"""Minimal stand-in for the `parameterized` package (absent from this image): `parameterized.expand` only."""
import inspect


class param:
    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs


class parameterized:
    param = param

    @staticmethod
    def expand(cases, name_func=None, **_):
        cases = list(cases)

        def deco(fn):
            ns = inspect.currentframe().f_back.f_locals
            for i, c in enumerate(cases):
                if isinstance(c, param):
                    a, k = c.args, c.kwargs
                elif isinstance(c, (list, tuple)):
                    a, k = tuple(c), {}
                else:
                    a, k = (c,), {}

                def make(a=a, k=k):
                    def t(self):
                        return fn(self, *a, **k)

                    t.__name__ = f"{fn.__name__}_{i}"
                    t.__doc__ = fn.__doc__
                    return t

                ns[f"{fn.__name__}_{i}"] = make()
            return None

        return deco
