registry = {}


def register(id, entry_point=None, **kwargs):
    registry[id] = dict(entry_point=entry_point, **kwargs)
