"""One generic way to turn a config node into a module: the reference repeats this logic in each of its builder
files (cost_processors/aggregators/builder.py:18-29, disp_predictors/builder.py:12-23, backbones/builder.py,
disp_refinement/builder.py:12-25); here every builder is a line of data plus a call to ``instantiate``."""


class UnknownType(KeyError):
    pass


def instantiate(table, node, what, default_type=None, off_path=(), **extra):
    """``table[node.type](**rest_of_node, **extra)``.

    ``node`` is a config dict (ConfigDict) holding ``type`` and the constructor's keyword arguments; ``extra`` are
    arguments the reference's builders inject from elsewhere in the config (``batch_norm``).  A type the reference
    knows but this path does not implement (``off_path``) raises NotImplementedError, anything else UnknownType."""
    kwargs = dict(node)
    kind = kwargs.pop("type", default_type)
    if kind in off_path:
        raise NotImplementedError("%s '%s' is outside the HIP path (see DESIGN.md, out of scope)" % (what, kind))
    if kind not in table:
        raise UnknownType("%s type not found, expected one of %s but got %r" % (what, sorted(table), kind))
    kwargs.update(extra)
    return table[kind](**kwargs)
