import json, sys
a = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
for k in a:
    ga, want = a[k]; gb, _ = b[k]
    if "weight" in k and "bn" not in k and "downsample.1" not in k:
        print(f"{k[37:]:40s} split {ga/want:6.3f}  unsplit {gb/want:6.3f}   (ref {want:.3g})")
