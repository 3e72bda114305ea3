"""Copies the compressed COCO RLE strings the reference ships in tests/data/vis_sample.json into
tests/golden/coco_rle_strings.json (with their `area` and `bbox`): the known answers that pin oracle/rle.py.
Run in the container that has /root/reference."""
import json
import os

SRC = '/root/reference/tests/data/vis_sample.json'
d = json.load(open(SRC))
items = []
for a in d['annotations']:
    seg = a['segmentation']
    if isinstance(seg, dict) and isinstance(seg.get('counts'), str):
        items.append(dict(size=seg['size'], counts=seg['counts'], area=a.get('area'), bbox=a.get('bbox')))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'coco_rle_strings.json')
json.dump(dict(source=SRC + ' (annotations[*].segmentation, compressed COCO RLE)', items=items), open(out, 'w'))
print(len(items), 'strings ->', out)
