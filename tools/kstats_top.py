import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot/1e6)
for r in rows[:14]:
    print("%5d %6.2f%% avg %8.1f  %s" % (int(r["Calls"]), float(r["Percentage"]), float(r["AverageNs"])/1e3, r["Name"][:95]))
