"""Minimal attribute-dict with the behaviour of the `easydict.EasyDict` the reference imports
(lib/lstm/config.py:5, lstm/train_net.py:14): nested dicts become attribute-accessible, and the strict
config merge relies on `type(x) is edict`."""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super(EasyDict, self).__init__()
        if d is None:
            d = {}
        if kwargs:
            d = dict(d, **kwargs)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if isinstance(value, (list, tuple)):
            value = type(value)(self.__class__(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x
                                for x in value)
        elif isinstance(value, dict) and not isinstance(value, EasyDict):
            value = self.__class__(value)
        super(EasyDict, self).__setitem__(name, value)

    __setitem__ = __setattr__

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __delattr__(self, name):
        del self[name]

    def update(self, e=None, **f):
        d = e or dict()
        d.update(f)
        for k in d:
            setattr(self, k, d[k])
