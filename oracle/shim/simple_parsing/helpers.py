import dataclasses
import typing
import warnings


def _build(tp, value):
    origin = typing.get_origin(tp)
    if origin is typing.Union or str(origin) == "<class 'types.UnionType'>":
        for a in typing.get_args(tp):
            if a is type(None):
                continue
            if dataclasses.is_dataclass(a) and isinstance(value, dict):
                return _from_dict(a, value)
        return value
    if dataclasses.is_dataclass(tp) and isinstance(value, dict):
        return _from_dict(tp, value)
    return value


def _from_dict(cls, d):
    hints = typing.get_type_hints(cls)
    names = {f.name for f in dataclasses.fields(cls)}
    kwargs = {}
    for k, v in d.items():
        if k not in names:
            warnings.warn(f"dropping unknown key {k!r} for {cls.__name__}")
            continue
        kwargs[k] = _build(hints[k], v) if v is not None else None
    return cls(**kwargs)


class Serializable:
    @classmethod
    def from_dict(cls, d, drop_extra_fields=None):
        return _from_dict(cls, dict(d))

    def to_dict(self):
        return dataclasses.asdict(self)
