"""ORACLE (test infrastructure only) — byte-at-a-time restatement of the reference's analysis window assembly.

Follows `internal/audiocore/buffer/analysis.go` statement by statement (Write :152-175, Read :187-251, Reset :266-272) over a
deliberately naive model of the third-party ring (`github.com/smallnest/ringbuffer v0.1.1`, go.mod:31, absent from the snapshot;
its published contract in overwrite mode: writes never fail, the oldest unread bytes are dropped first).  Pure-Python loops:
small cases only.  Pinned by the reference's own cases (analysis_test.go:23-243) in tests/test_stream.py.
"""


class GoRing:
    """A list of unread bytes with a capacity: the slowest possible statement of an overwriting byte ring."""

    def __init__(self, capacity):
        self.cap, self.q = capacity, []

    def Length(self):
        return len(self.q)

    def Free(self):
        return self.cap - len(self.q)

    def Write(self, p):
        for b in p:
            if len(self.q) == self.cap:
                self.q.pop(0)
            self.q.append(b)
        return len(p)

    def Read(self, n):
        out, self.q = self.q[:n], self.q[n:]
        return out

    def Reset(self):
        self.q = []


class GoAnalysisBuffer:
    def __init__(self, capacity, overlapSize, readSize):
        self.ring = GoRing(capacity)
        self.prevData = None
        self.overlapSize, self.readSize, self.windowSize = overlapSize, readSize, overlapSize + readSize
        self.overwrites = 0

    def Write(self, data):                                   # analysis.go:152-175
        willOverwrite = len(data) > self.ring.Free()
        self.ring.Write(data)
        if willOverwrite:
            self.overwrites += 1

    def Read(self):                                          # analysis.go:187-251
        if self.ring.Length() < self.readSize:
            return None
        window = [0xEE] * self.windowSize                    # a pooled slice holds stale bytes: poison, every byte must be set
        if self.overlapSize > 0:
            if self.prevData is not None and len(self.prevData) == self.overlapSize:
                for i in range(self.overlapSize):
                    window[i] = self.prevData[i]
            else:
                for i in range(self.overlapSize):
                    window[i] = 0
        got = self.ring.Read(self.readSize)
        n = len(got)
        for i in range(n):
            window[self.overlapSize + i] = got[i]
        if n < self.readSize:
            for i in range(self.overlapSize + n, self.windowSize):
                window[i] = 0
        if self.overlapSize > 0:
            if self.prevData is None:
                self.prevData = [0] * self.overlapSize
            freshEnd = self.overlapSize + n
            if n >= self.overlapSize:
                self.prevData = window[freshEnd - self.overlapSize:freshEnd]
            else:
                self.prevData = [0] * self.overlapSize
                for i in range(n):
                    self.prevData[self.overlapSize - n + i] = window[self.overlapSize + i]
        return bytes(window)

    def Reset(self):                                         # analysis.go:266-272
        self.ring.Reset()
        self.prevData = None
