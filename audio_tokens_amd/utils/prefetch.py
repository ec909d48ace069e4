"""Runs a generator one step ahead in a background thread, so that reading the next batch of .npy
files (the reference's inter-stage protocol, cluster_creator.py:83-102, spec_tokenizer.py:66-69)
overlaps the GPU work on the current one.  np.load releases the GIL while it reads."""
import queue
import threading

_END = object()


def prefetch(generator, depth: int = 1):
    """Yields exactly what `generator` yields, in order; exceptions raised by it are re-raised here."""
    q = queue.Queue(maxsize=max(1, depth))

    def produce():
        try:
            for item in generator:
                q.put((item, None))
            q.put((_END, None))
        except BaseException as e:  # noqa: BLE001 - handed to the consumer
            q.put((_END, e))

    t = threading.Thread(target=produce, name="audio-tokens-prefetch", daemon=True)
    t.start()
    while True:
        item, err = q.get()
        if item is _END:
            t.join()
            if err is not None:
                raise err
            return
        yield item
