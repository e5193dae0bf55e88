import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].replace("dabk::", "")
    acc[(r.get("Queue_Id", "?"), k)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
qs = sorted(set(q for q, _ in acc))
ks = ["copy_pieces_kernel", "prs_sync_kernel", "ofdm_wave_kernel", "track_update_kernel", "viterbi_rot_grouped_kernel"]
print("queue".ljust(8) + "".join(k[:22].rjust(24) for k in ks))
for q in qs:
    print(str(q).ljust(8) + "".join((("%.1f us (%d)" % (sum(acc[(q, k)][-60:]) / len(acc[(q, k)][-60:]), len(acc[(q, k)]))) if acc.get((q, k)) else "-").rjust(24) for k in ks))
