def to_type_converter(spec):
    return spec
