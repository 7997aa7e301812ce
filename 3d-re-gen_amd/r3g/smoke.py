"""Extra smoke checks appended as components land (called from __graft_entry__.smoke())."""


def run():
    return None
