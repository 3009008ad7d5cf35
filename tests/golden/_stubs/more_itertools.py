"""Stand-in for more_itertools (only `grouper`, used at reference src/ptwt/_util.py:804)."""


def grouper(iterable, n):
    it = list(iterable)
    return [tuple(it[i : i + n]) for i in range(0, len(it), n)]
